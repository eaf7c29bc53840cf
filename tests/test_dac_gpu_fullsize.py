"""Size-independent properties of the DAC path at BASELINE.json configs[2]'s clip shape (DAC 44.1 kHz,
9 codebooks, 10 s clips); 48 clips so that the batch crosses the internal clip-chunk boundary (39 clips at
the 40 GB workspace cap)."""
import numpy as np
import pytest
import torch

from golden_cases import noise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec(dac_checkpoints):
    from audiocodecs_amd import DAC

    cfg, sd = dac_checkpoints("full", 0)
    return DAC(44100, 44100, num_codebooks=9, state_dict=sd, config=cfg).eval()


def test_full_batch_properties(codec, dac_checkpoints):
    from oracle import dac_oracle as O
    from test_oracle_golden import tokens_match_up_to_ties

    B, T = 48, 441000
    sig = noise(654, B, T).cuda()
    toks = codec.sig_to_toks(sig)
    assert toks.shape == (B, 861, 9) and toks.dtype == torch.int64
    assert int(toks.min()) >= 0 and int(toks.max()) < 1024
    assert torch.equal(toks, codec.sig_to_toks(sig))                       # deterministic
    for b in (0, 38, 39, 47):                                              # both sides of the chunk boundary
        assert torch.equal(codec.sig_to_toks(sig[b : b + 1]), toks[b : b + 1])
    rec = codec.toks_to_sig(toks)
    assert rec.shape == (B, 861 * 512) and bool(torch.isfinite(rec).all())
    assert float(rec.abs().max()) <= 1.0                                   # tanh head
    for b in (5, 39, 47):
        assert torch.equal(codec.toks_to_sig(toks[b : b + 1]), rec[b : b + 1])
    # spot-check one whole clip (second chunk) against the CPU oracle
    cfg, sd = dac_checkpoints("full", 0)
    W, W64 = O.cast_weights(sd), O.cast_weights(sd, torch.float64)
    idx = [41]
    with torch.no_grad():
        s = sig[idx].cpu()
        otoks = O.sig_to_toks(cfg, W, s, None, 9)
        _, m64 = O.sig_to_toks(cfg, W64, s.double(), None, 9, "descript", True)
        orec = O.toks_to_sig(cfg, W, otoks)
    import parity_record
    from test_oracle_golden import TAU

    mism, bad, excused = parity_record.tokens("dac", "fullsize_spot_check", toks[idx].cpu().numpy(), otoks.numpy(), m64.numpy(), TAU)
    assert bad == 0 and mism <= excused
    err = (codec.toks_to_sig(otoks.cuda()).cpu() - orec).numpy().astype(np.float64)
    assert float(np.sqrt(np.mean(err**2))) < 3e-5


def test_config3_as_stated_256_clips(codec):
    """BASELINE.json configs[2] at its stated size: 256 clips x 10 s in ONE call = 7 chunks of <= 39 clips through the 40 GB
    workspace cap.  Every chunk starts a fresh pass over the workspace's amax pool (csrc/ac_api.hip dac_encode_impl /
    ac_decode: amax_begin per chunk), so the clips on both sides of EVERY chunk boundary -- and the first and last clip of the
    batch -- must be bit-equal to the same clip run alone, encode and decode; reruns of the whole batch repeat bit for bit."""
    B, T = 256, 441000
    g = torch.Generator(device="cuda").manual_seed(31)
    sig = 0.1 * torch.randn(B, T, generator=g, device="cuda")              # 452 MB; drawn on the device
    toks = codec.sig_to_toks(sig)
    assert toks.shape == (B, 861, 9) and int(toks.min()) >= 0 and int(toks.max()) < 1024
    chunk = 39                                                             # dac_chunk_clips at 10 s clips (asserted via the boundaries below)
    edges = sorted({0, B - 1} | {c for k in range(1, B // chunk + 1) for c in (k * chunk - 1, k * chunk) if c < B})
    assert len(edges) == 14
    for b in edges:
        assert torch.equal(codec.sig_to_toks(sig[b : b + 1]), toks[b : b + 1]), b
    assert torch.equal(codec.sig_to_toks(sig), toks)                       # rerun of all 7 chunks
    rec = codec.toks_to_sig(toks)
    assert rec.shape == (B, 861 * 512) and bool(torch.isfinite(rec).all()) and float(rec.abs().max()) <= 1.0
    for b in edges:
        assert torch.equal(codec.toks_to_sig(toks[b : b + 1]), rec[b : b + 1]), b
    assert torch.equal(codec.toks_to_sig(toks), rec)
    assert len({tuple(t.flatten().tolist()[:64]) for t in toks[::37]}) > 1  # different clips, different tokens (no chunk was reused)
