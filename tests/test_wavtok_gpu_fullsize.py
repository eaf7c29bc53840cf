"""Size-independent properties at BASELINE.json configs[4]'s per-GPU size (WavTokenizer 40 tok/s, 64 clips x 10 s) where
the CPU oracle is too slow to check the whole batch; one whole clip is spot-checked against it."""
import numpy as np
import pytest
import torch

from golden_cases import noise

pytestmark = pytest.mark.gpu


def test_full_batch_properties(wavtok_checkpoints):
    import parity_record
    from audiocodecs_amd import WavTokenizer
    from oracle import wavtokenizer_oracle as O
    from test_oracle_golden import TAU

    cfg, sd = wavtok_checkpoints("full", 0)
    codec = WavTokenizer(24000, state_dict=sd, arch=cfg).eval()
    B, T = 64, 240000
    sig = noise(523, B, T).cuda()
    toks = codec.sig_to_toks(sig)
    assert toks.shape == (B, 400, 1) and toks.dtype == torch.int64
    assert int(toks.min()) >= 0 and int(toks.max()) < 4096
    assert torch.equal(toks, codec.sig_to_toks(sig))                       # run-to-run determinism
    for b in (0, 37, 63):                                                   # clips are independent units
        assert torch.equal(codec.sig_to_toks(sig[b : b + 1]), toks[b : b + 1])
    rec = codec.toks_to_sig(toks)
    assert rec.shape == (B, T) and bool(torch.isfinite(rec).all())
    assert torch.equal(codec.toks_to_sig(toks[5:6]), rec[5:6])
    assert torch.equal(rec, codec.toks_to_sig(toks))
    # FLOAT outputs repeat bit for bit (round-3 advisor: the packed-FMA fault of the EnCodec front was visible only in such a test)
    feats = codec.sig_to_feats(sig)
    for _ in range(3):
        assert torch.equal(codec.sig_to_feats(sig), feats)
        assert torch.equal(codec.toks_to_sig(toks), rec)
    nat = next(iter(codec._natives.values()))
    assert nat.lib.ac_lstm_status(nat.h) == 1                               # the persistent LSTM ran, no failed launch
    W, W64 = O.cast_weights(sd), O.cast_weights(sd, torch.float64)
    idx = [41]
    with torch.no_grad():
        s = sig[idx].cpu()
        otoks = O.sig_to_toks(cfg, W, s)
        _, m64 = O.sig_to_toks(cfg, W64, s.double(), True)
        orec = O.toks_to_sig(cfg, W, otoks)
    mism, bad, excused = parity_record.tokens("wavtokenizer", "fullsize_spot_check", toks[idx].cpu().numpy(), otoks.numpy(), m64.numpy(), TAU)
    assert bad == 0 and mism <= excused
    err = (codec.toks_to_sig(otoks.cuda()).cpu() - orec).numpy().astype(np.float64)
    e = float(np.sqrt(np.mean(err**2)))
    parity_record.record("wavtokenizer", "fullsize_spot_check", waveform_rms_err=e)
    assert e < 2e-5
