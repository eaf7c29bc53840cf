"""The OPT-IN bf16 mode (ac_set_precision(AC_PRECISION_BF16)): the GEMM-shaped kernels
(tap-GEMMs, fused residual blocks, [64][128] layers) round their operands to
bf16 and do one product per pair.  It is NOT a parity mode -- token ids differ from the reference wherever the codebook
margin is below the bf16 noise -- so this test only checks that the mode runs, stays close to the fp32-faithful default in
the signal domain, and RECORDS its own mismatch rate and errors (parity_report.json), as the survey asks."""
import numpy as np
import pytest
import torch

import parity_record
from golden_cases import noise

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a, b = a.double().cpu().numpy(), b.double().cpu().numpy()
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(1e-30, np.sqrt(np.mean(b**2))))


def test_encodec_bf16_mode_is_close_and_reported(checkpoints):
    from audiocodecs_amd import Encodec

    cfg, sd = checkpoints("full", 0)
    ref = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    low = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg, precision="bf16").eval()
    sig = noise(7171, 4, 48000).cuda()
    tr, tl = ref.sig_to_toks(sig), low.sig_to_toks(sig)
    fr, fl = ref.sig_to_feats(sig), low.sig_to_feats(sig)
    rr, rl = ref.toks_to_sig(tr), low.toks_to_sig(tr)          # same tokens: isolates the decoder's arithmetic
    match_all = float((tr == tl).float().mean())
    match_first = float((tr[..., 0] == tl[..., 0]).float().mean())
    parity_record.record("encodec_bf16_mode", "noise_b4_2s", token_match_all_stages=match_all, token_match_first_stage=match_first,
                         feats_rel_rms=rel_rms(fl, fr), decode_rel_rms=rel_rms(rl, rr), note="opt-in side mode; not a parity claim")
    assert rel_rms(fl, fr) < 0.05 and rel_rms(rl, rr) < 0.05
    assert match_first > 0.5                                   # the first stage has the widest margins
    assert torch.equal(low.sig_to_toks(sig), tl)               # deterministic
    with pytest.raises(ValueError):
        Encodec(24000, state_dict=sd, config=cfg, precision="fp8")


def test_wavtokenizer_bf16_mode_runs(wavtok_checkpoints):
    from audiocodecs_amd import WavTokenizer

    cfg, sd = wavtok_checkpoints("full", 0)
    ref = WavTokenizer(24000, state_dict=sd, arch=cfg).eval()
    low = WavTokenizer(24000, state_dict=sd, arch=cfg, precision="bf16").eval()
    sig = noise(7272, 3, 36000).cuda()
    tr, tl = ref.sig_to_toks(sig), low.sig_to_toks(sig)
    rr, rl = ref.toks_to_sig(tr), low.toks_to_sig(tr)
    parity_record.record("wavtokenizer_bf16_mode", "noise_b3_1.5s", token_match_all_stages=float((tr == tl).float().mean()),
                         decode_rel_rms=rel_rms(rl, rr), note="opt-in side mode; not a parity claim")
    assert rel_rms(rl, rr) < 0.1
