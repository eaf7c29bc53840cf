"""Sample-rate conversion at the Codec boundary.  Parity with torchaudio is
unpinned (not on disk); these tests pin the product to the fp64 restatement in oracle/ and to the
analytic answer on band-limited tones."""
import math

import numpy as np
import pytest
import torch

from audiocodecs_amd.resample import sinc_kernel
from oracle import resample_oracle as R


@pytest.mark.parametrize("rates,shape,width", [((16000, 24000), (3, 16), 7), ((24000, 16000), (2, 23), 10), ((16000, 44100), (441, 174), 7)])
def test_filter_bank_shapes_and_values(rates, shape, width):
    k, n, o, w = sinc_kernel(*rates)
    assert tuple(k.shape) == shape and w == width and n == shape[0]
    k64, n64, o64, w64 = R.kernel(*rates)
    np.testing.assert_allclose(k.numpy(), k64, atol=3e-5)   # the bank is evaluated in fp32 like torchaudio does
    assert abs(float(k.sum(1).mean()) - 1.0) < 2e-3      # unit DC gain per phase


def _tones(L, sr, freqs=(440.0, 2310.0), amps=(0.3, 0.2)):
    t = np.arange(L) / sr
    return sum(a * np.sin(2 * np.pi * f * t + 0.3 * i) for i, (f, a) in enumerate(zip(freqs, amps)))


def test_oracle_resamples_band_limited_tones():
    x = _tones(4000, 16000)[None]
    y = R.resample(x, 16000, 24000)
    assert y.shape == (1, 6000)
    ref = _tones(6000, 24000)
    assert np.abs(y[0, 200:-200] - ref[200:-200]).max() < 2e-3
    assert R.resample(x, 16000, 16000) is x
    assert R.resample(x[:, :1001], 24000, 16000).shape == (1, math.ceil(2 * 1001 / 3))


@pytest.mark.gpu
@pytest.mark.parametrize("rates", [(16000, 24000), (24000, 16000), (16000, 44100), (44100, 16000)])
def test_hip_resample_matches_oracle(rates):
    from audiocodecs_amd.resample import resample
    from golden_cases import noise

    x = noise(71, 3, 5003, amp=0.3)
    y = resample(x.cuda(), *rates).cpu().numpy()
    ref = R.resample(x.numpy(), *rates)
    assert y.shape == ref.shape
    # fp32 filter bank (as torchaudio evaluates it) + fp32 accumulation vs the fp64 restatement
    assert np.abs(y - ref).max() < (2e-6 if max(rates) == 24000 else 3e-5)


@pytest.mark.gpu
def test_readme_quickstart_flow_16k(checkpoints):
    """README.md:69-80 / BASELINE.json configs[0]: example.wav (16 kHz) through a 24 kHz codec and back:
    253760 -> 380640 samples -> 1190 frames -> 380800 -> 253867 samples."""
    from audiocodecs_amd import Encodec
    from conftest import GOLDEN_DIR
    from golden_cases import read_example_wav

    cfg, sd = checkpoints("full", 0)
    codec = Encodec(16000, orig_sample_rate=24000, num_codebooks=8, state_dict=sd).eval()
    sig = read_example_wav(GOLDEN_DIR).cuda()
    toks = codec.sig_to_toks(sig)
    assert toks.shape == (1, 1190, 8)
    rec = codec(sig)
    assert rec.shape == (1, 253867) and bool(torch.isfinite(rec).all())
    assert torch.equal(rec, codec.toks_to_sig(toks))
