"""Size-independent properties of the Mimi path at BASELINE.json configs[3]'s per-GPU shape
(Mimi 24 kHz, 8 codebooks, 128 clips x 10 s on one MI355X = 1024 clips over 8 GPUs)."""
import numpy as np
import pytest
import torch

from golden_cases import noise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec(mimi_checkpoints):
    from audiocodecs_amd import Mimi

    cfg, sd = mimi_checkpoints("full", 0)
    return Mimi(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()


def test_full_batch_properties(codec, mimi_checkpoints):
    from oracle import mimi_oracle as O
    from test_oracle_golden import tokens_match_up_to_ties

    B, T = 128, 240000
    sig = noise(321, B, T).cuda()
    toks = codec.sig_to_toks(sig)
    assert toks.shape == (B, 125, 8) and toks.dtype == torch.int64
    assert int(toks.min()) >= 0 and int(toks.max()) < 2048
    assert torch.equal(toks, codec.sig_to_toks(sig))                      # deterministic
    for b in (0, 77, 127):                                                # clips are independent units
        assert torch.equal(codec.sig_to_toks(sig[b : b + 1]), toks[b : b + 1])
    rec = codec.toks_to_sig(toks)
    assert rec.shape == (B, T) and bool(torch.isfinite(rec).all())
    assert torch.equal(codec.toks_to_sig(toks[5:6]), rec[5:6])
    # causality (causal convs, causal attention): the first 4.8 s of tokens do not depend on the rest
    head = codec.sig_to_toks(sig[:2, : 60 * 1920])
    assert torch.equal(head, toks[:2, :60])
    # spot-check one whole clip against the CPU oracle
    cfg, sd = mimi_checkpoints("full", 0)
    W, W64 = O.cast_weights(sd), O.cast_weights(sd, torch.float64)
    idx = [101]
    with torch.no_grad():
        s = sig[idx].cpu()
        otoks = O.sig_to_toks(cfg, W, s)
        _, m64 = O.sig_to_toks(cfg, W64, s.double(), None, 8, True)
        orec = O.toks_to_sig(cfg, W, otoks)
    import parity_record
    from test_oracle_golden import TAU

    mism, bad, excused = parity_record.tokens("mimi", "fullsize_spot_check", toks[idx].cpu().numpy(), otoks.numpy(), m64.numpy(), TAU)
    assert bad == 0 and mism <= excused
    err = (codec.toks_to_sig(otoks.cuda()).cpu() - orec).numpy().astype(np.float64)
    assert float(np.sqrt(np.mean(err**2))) < 2e-5
