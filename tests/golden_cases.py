"""Case list for the golden fixtures (shared by tools/make_golden.py and the tests).

Inputs are never stored: both sides re-draw them from the repo PRNG (audiocodecs_amd.prng), so a
fixture only carries what the reference produced.  The single exception is example.wav -- the
reference's own data file (audiocodecs/example.wav, mono PCM16 16 kHz, 253 760 samples), read with
the stdlib `wave` module and scaled by 1/32768 exactly like `torchaudio.load` normalises PCM16.
Fed at the codec's native rate (sample_rate == orig_sample_rate == 24000) so that no resampler --
whose torchaudio implementation is not on disk -- sits between the fixture and the path.
"""

from __future__ import annotations

import os
import wave

import numpy as np
import torch

from audiocodecs_amd import prng

REC_STRIDE = 61  # golden waveforms/features are stored as every 61st element (+ RMS + sha256)

CASES = [
    # full EnCodec-24k architecture, weights seed 0
    dict(name="full_example", cfg="full", weights_seed=0, kind="wav"),
    dict(name="full_noise_b2", cfg="full", weights_seed=0, kind="noise", B=2, T=24000, seed=11),
    dict(name="full_ragged_b3", cfg="full", weights_seed=0, kind="noise", B=3, T=24001, seed=12,
         length=[1.0, 0.7, 0.31]),
    dict(name="full_T1", cfg="full", weights_seed=0, kind="noise", B=2, T=1, seed=13),
    dict(name="full_T5", cfg="full", weights_seed=0, kind="noise", B=1, T=5, seed=19),
    dict(name="full_T319", cfg="full", weights_seed=0, kind="noise", B=2, T=319, seed=14),
    dict(name="full_T320", cfg="full", weights_seed=0, kind="noise", B=2, T=320, seed=15),
    dict(name="full_T321", cfg="full", weights_seed=0, kind="noise", B=2, T=321, seed=16),
    dict(name="full_T2477_K2", cfg="full", weights_seed=0, kind="noise", B=1, T=2477, seed=17, K=2),
    dict(name="full_T2477_K32", cfg="full", weights_seed=0, kind="noise", B=1, T=2477, seed=17, K=32),
    dict(name="full_tones_b2", cfg="full", weights_seed=0, kind="tones", B=2, T=12000, seed=18),
    dict(name="full_decode_rand", cfg="full", weights_seed=0, kind="decode", B=2, N=10, K=8, seed=21),
    dict(name="full_decode_K16", cfg="full", weights_seed=0, kind="decode", B=1, N=7, K=16, seed=22),
    dict(name="full_w1_noise", cfg="full", weights_seed=1, kind="noise", B=1, T=4800, seed=31),
    # tiny architecture (num_filters=4, hidden_size=16): every module output is in the fixture
    dict(name="tiny_taps", cfg="tiny", weights_seed=0, kind="noise", B=2, T=1000, seed=41, taps=True),
    dict(name="tiny_ragged", cfg="tiny", weights_seed=0, kind="noise", B=2, T=777, seed=42,
         length=[1.0, 0.5], taps=True),
]


def read_example_wav(golden_dir: str) -> torch.Tensor:
    with wave.open(os.path.join(golden_dir, "example.wav")) as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return torch.from_numpy((pcm.astype(np.float32) / 32768.0)[None])


def noise(seed: int, B: int, T: int, amp: float = 0.1) -> torch.Tensor:
    return torch.from_numpy((prng.normal(seed, "sig", (B, T)) * amp).astype(np.float32))


def _sin2pi(p: np.ndarray) -> np.ndarray:
    """sin(2*pi*p) from +,-,* only (odd Taylor polynomial after quadrant reduction), so every host
    computes the same bits -- libm's sin is not guaranteed to."""
    p = p - np.floor(p)
    q = np.where(p > 0.75, p - 1.0, np.where(p > 0.25, 0.5 - p, p))  # sin(2 pi p) == sin(2 pi q), |q| <= 1/4
    x = q * 6.283185307179586
    x2 = x * x
    acc = np.full_like(x, -1.0 / 1307674368000.0)  # -x^15/15!
    for c in (1.0 / 6227020800.0, -1.0 / 39916800.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0, -1.0 / 6.0, 1.0):
        acc = acc * x2 + c
    return acc * x


def tones(seed: int, B: int, T: int) -> torch.Tensor:
    """Speech-like: amplitude-modulated multi-sine (float64 arithmetic, rounded once to float32)."""
    t = np.arange(T, dtype=np.float64) * (1.0 / 24000.0)
    f = prng.uniform(seed, "f", (B, 5), 80.0, 3000.0)
    a = prng.uniform(seed, "a", (B, 5), 0.02, 0.12)
    x = (a[:, :, None] * _sin2pi(f[:, :, None] * t[None, None])).sum(1)
    env = 0.55 + 0.45 * _sin2pi(3.0 * t)[None]
    return torch.from_numpy((x * env).astype(np.float32))


def make_input(case: dict, golden_dir: str) -> dict:
    kind = case["kind"]
    if kind == "wav":
        return {"sig": read_example_wav(golden_dir)}
    if kind == "noise":
        out = {"sig": noise(case["seed"], case["B"], case["T"])}
    elif kind == "tones":
        out = {"sig": tones(case["seed"], case["B"], case["T"])}
    elif kind == "decode":
        return {"toks": torch.from_numpy(prng.randint(case["seed"], "toks", (case["B"], case["N"], case["K"]), 1024))}
    else:
        raise ValueError(kind)
    if "length" in case:
        out["length"] = torch.tensor(case["length"], dtype=torch.float32)
    return out
