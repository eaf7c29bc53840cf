"""Shape sweep: seeded (B, T) combinations that exercise partial tiles of every tile arrangement (128 / 256-column tiles, 1 x 8
waves, 64 x 96), partial 16-clip LSTM groups, clips shorter than the padding, and the non-causal edges of WavTokenizer --
HIP path vs the CPU oracle with the usual bars (token ids exact outside fp64 near-ties, waveform RMS < 1e-5 / 2e-5)."""
import numpy as np
import pytest
import torch

import parity_record
from golden_cases import noise
from test_oracle_golden import TAU

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1), (2, 7), (1, 333), (3, 641), (7, 1283), (2, 4801), (17, 2560), (5, 9999), (1, 31999), (2, 24000)]


def rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, dtype=np.float64) ** 2)))


@pytest.mark.parametrize("B,T", SHAPES)
def test_encodec(B, T, checkpoints):
    from audiocodecs_amd import Encodec
    from oracle import encodec_oracle as O

    cfg, sd = checkpoints("full", 0)
    codec = _cached("encodec", lambda: Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval())
    W, W64 = O.fold_weight_norm(sd), O.fold_weight_norm(sd, torch.float64)
    sig = noise(10000 + 7 * B + T, B, T)
    with torch.no_grad():
        otoks = O.sig_to_toks(cfg, W, sig)
        _, m64 = O.sig_to_toks(cfg, W64, sig.double(), None, 8, True)
        orec = O.toks_to_sig(cfg, W, otoks)
    toks = codec.sig_to_toks(sig.cuda())
    mism, bad, excused = parity_record.tokens("encodec", f"sweep_B{B}_T{T}", toks.cpu().numpy(), otoks.numpy(), m64.numpy(), TAU)
    assert bad == 0 and mism <= excused
    assert rms((codec.toks_to_sig(otoks.cuda()).cpu() - orec).numpy()) < 1e-5


@pytest.mark.parametrize("B,T", SHAPES)
@pytest.mark.parametrize("arch", ["full", "f75"])
def test_wavtokenizer(arch, B, T, wavtok_checkpoints):
    from audiocodecs_amd import WavTokenizer
    from oracle import wavtokenizer_oracle as O

    cfg, sd = wavtok_checkpoints(arch, 0)
    codec = _cached("wavtok_" + arch, lambda: WavTokenizer(24000, state_dict=sd, arch=cfg).eval())
    W, W64 = O.cast_weights(sd), O.cast_weights(sd, torch.float64)
    sig = noise(20000 + 7 * B + T, B, T)
    with torch.no_grad():
        otoks = O.sig_to_toks(cfg, W, sig)
        _, m64 = O.sig_to_toks(cfg, W64, sig.double(), True)
        orec = O.toks_to_sig(cfg, W, otoks)
    toks = codec.sig_to_toks(sig.cuda())
    mism, bad, excused = parity_record.tokens("wavtokenizer", f"sweep_{arch}_B{B}_T{T}", toks.cpu().numpy(), otoks.numpy(), m64.numpy(), TAU)
    assert bad == 0 and mism <= excused
    assert rms((codec.toks_to_sig(otoks.cuda()).cpu() - orec).numpy()) < 2e-5


_CACHE = {}


def _cached(key, make):
    if key not in _CACHE:
        _CACHE[key] = make()
    return _CACHE[key]


MIMI_SHAPES = [(1, 1), (3, 1921), (2, 5000), (7, 3841), (17, 2000)]
DAC_SHAPES = [(1, 512), (3, 777), (2, 2049), (5, 1300)]


@pytest.mark.parametrize("B,T", MIMI_SHAPES)
def test_mimi(B, T, mimi_checkpoints):
    from audiocodecs_amd import Mimi
    from oracle import mimi_oracle as O

    cfg, sd = mimi_checkpoints("full", 0)
    codec = _cached("mimi", lambda: Mimi(24000, num_codebooks=8, state_dict=sd, config=cfg).eval())
    W, W64 = O.cast_weights(sd), O.cast_weights(sd, torch.float64)
    sig = noise(30000 + 7 * B + T, B, T)
    with torch.no_grad():
        otoks = O.sig_to_toks(cfg, W, sig)
        _, m64 = O.sig_to_toks(cfg, W64, sig.double(), None, 8, True)
        orec = O.toks_to_sig(cfg, W, otoks)
    toks = codec.sig_to_toks(sig.cuda())
    mism, bad, excused = parity_record.tokens("mimi", f"sweep_B{B}_T{T}", toks.cpu().numpy(), otoks.numpy(), m64.numpy(), TAU)
    assert bad == 0 and mism <= excused
    assert rms((codec.toks_to_sig(otoks.cuda()).cpu() - orec).numpy()) < 2e-5


@pytest.mark.parametrize("B,T", DAC_SHAPES)
def test_dac(B, T, dac_checkpoints):
    from audiocodecs_amd import DAC
    from oracle import dac_oracle as O

    cfg, sd = dac_checkpoints("full", 0)
    codec = _cached("dac", lambda: DAC(44100, 44100, num_codebooks=9, state_dict=sd, config=cfg).eval())
    W, W64 = O.cast_weights(sd), O.cast_weights(sd, torch.float64)
    sig = noise(40000 + 7 * B + T, B, T)
    with torch.no_grad():
        otoks = O.sig_to_toks(cfg, W, sig, None, 9)
        _, m64 = O.sig_to_toks(cfg, W64, sig.double(), None, 9, return_margin=True)
        orec = O.toks_to_sig(cfg, W, otoks)
    toks = codec.sig_to_toks(sig.cuda())
    mism, bad, excused = parity_record.tokens("dac", f"sweep_B{B}_T{T}", toks.cpu().numpy(), otoks.numpy(), m64.numpy(), TAU)
    assert bad == 0 and mism <= excused
    assert rms((codec.toks_to_sig(otoks.cuda()).cpu() - orec).numpy()) < 3e-5
