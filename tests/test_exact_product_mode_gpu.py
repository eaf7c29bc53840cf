"""Exact-fp32-product kernels (precision="fp32_exact" == AC_GEMM=fp32: tap_gemm4, rb_fused, lstm_persist) against the
default split-operand kernels for Mimi, DAC and WavTokenizer (EnCodec: tests/test_gpu_parity.py).  Same function up to
fp32-level rounding; the exact-product build also reproduces each codec's fixture tokens outside fp64 near-ties."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from golden_cases import noise
from test_oracle_golden import TAU, tokens_match_up_to_ties

pytestmark = pytest.mark.gpu


def rms(a):
    return float(np.sqrt(np.mean(np.asarray(a.detach().cpu().numpy(), dtype=np.float64) ** 2)))


def _check(fast, exact, sig, z, cases_mod, tol):
    fa, fb = fast.sig_to_feats(sig), exact.sig_to_feats(sig)
    assert rms(fa - fb) < tol * max(1.0, rms(fb))
    ta, tb = fast.sig_to_toks(sig), exact.sig_to_toks(sig)
    assert float((ta == tb).float().mean()) > 0.995
    ra, rb = fast.toks_to_sig(ta), exact.toks_to_sig(ta)
    assert rms(ra - rb) < tol * max(1.0, rms(rb))
    case = next(c for c in cases_mod.CASES if c["name"] == "full_noise_b2")
    inp = cases_mod.make_input(case, GOLDEN_DIR)
    toks = exact.sig_to_toks(inp["sig"].cuda())
    n, bad, excused = tokens_match_up_to_ties(toks.cpu().numpy(), z["full_noise_b2.toks"].astype(np.int64), z["full_noise_b2.margin64"])
    assert bad == 0


def test_mimi(mimi_checkpoints, mimi_golden):
    import mimi_cases
    from audiocodecs_amd import Mimi

    cfg, sd = mimi_checkpoints("full", 0)
    fast = Mimi(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    exact = Mimi(24000, num_codebooks=8, state_dict=sd, config=cfg, precision="fp32_exact").eval()
    _check(fast, exact, noise(6001, 2, 48000).cuda(), mimi_golden[0], mimi_cases, 1e-5)


def test_dac(dac_checkpoints, dac_golden):
    import dac_cases
    from audiocodecs_amd import DAC

    cfg, sd = dac_checkpoints("full", 0)
    fast = DAC(44100, 44100, num_codebooks=9, state_dict=sd, config=cfg).eval()
    exact = DAC(44100, 44100, num_codebooks=9, state_dict=sd, config=cfg, precision="fp32_exact").eval()
    _check(fast, exact, noise(6002, 2, 30000).cuda(), dac_golden[0], dac_cases, 2e-5)


def test_wavtokenizer(wavtok_checkpoints, wavtok_golden):
    import wavtok_cases
    from audiocodecs_amd import WavTokenizer

    cfg, sd = wavtok_checkpoints("full", 0)
    fast = WavTokenizer(24000, state_dict=sd, arch=cfg).eval()
    exact = WavTokenizer(24000, state_dict=sd, arch=cfg, precision="fp32_exact").eval()
    _check(fast, exact, noise(6003, 3, 30000).cuda(), wavtok_golden[0], wavtok_cases, 1e-5)
