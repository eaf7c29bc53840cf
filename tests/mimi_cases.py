"""Case list for the Mimi golden fixtures (shared by tools/make_golden_mimi.py and the tests).

Same rules as golden_cases.py: inputs are re-drawn from the repo PRNG on both sides; only what the
reference wrapper (audiocodecs/mimi.py on top of transformers' MimiModel with OUR seeded synthetic
weights) produced is stored.  hop = 8*6*5*4*2 = 1920 samples; vocabulary 2048.
"""

from __future__ import annotations

import torch

from audiocodecs_amd import prng
from golden_cases import noise, read_example_wav, tones

REC_STRIDE = 61

CASES = [
    # full Mimi architecture (kyutai/mimi config), weights seed 0
    dict(name="full_example", cfg="full", weights_seed=0, kind="wav"),  # 265 frames @25 Hz > sliding_window 250
    dict(name="full_noise_b2", cfg="full", weights_seed=0, kind="noise", B=2, T=24000, seed=111),
    dict(name="full_ragged_b2", cfg="full", weights_seed=0, kind="noise", B=2, T=9601, seed=112, length=[1.0, 0.4]),
    dict(name="full_T1", cfg="full", weights_seed=0, kind="noise", B=2, T=1, seed=113),
    dict(name="full_T1919", cfg="full", weights_seed=0, kind="noise", B=1, T=1919, seed=114),
    dict(name="full_T1920", cfg="full", weights_seed=0, kind="noise", B=1, T=1920, seed=115),
    dict(name="full_T1921", cfg="full", weights_seed=0, kind="noise", B=1, T=1921, seed=116),
    dict(name="full_T4800_K1", cfg="full", weights_seed=0, kind="noise", B=1, T=4800, seed=117, K=1),
    dict(name="full_T4800_K32", cfg="full", weights_seed=0, kind="noise", B=1, T=4800, seed=117, K=32),
    dict(name="full_tones_b2", cfg="full", weights_seed=0, kind="tones", B=2, T=12000, seed=118),
    dict(name="full_decode_rand", cfg="full", weights_seed=0, kind="decode", B=2, N=10, K=8, seed=121),
    dict(name="full_decode_K1", cfg="full", weights_seed=0, kind="decode", B=1, N=3, K=1, seed=122),
    dict(name="full_w1_noise", cfg="full", weights_seed=1, kind="noise", B=1, T=7000, seed=131),
    # tiny architecture (1/8 width, 2 transformer layers, sliding_window 6): every module output stored
    dict(name="tiny_taps", cfg="tiny", weights_seed=0, kind="noise", B=2, T=9600, seed=141, taps=True),
    dict(name="tiny_odd", cfg="tiny", weights_seed=0, kind="noise", B=3, T=5555, seed=142, taps=True, K=5),
]


def make_input(case: dict, golden_dir: str) -> dict:
    kind = case["kind"]
    if kind == "wav":
        return {"sig": read_example_wav(golden_dir)}
    if kind == "noise":
        out = {"sig": noise(case["seed"], case["B"], case["T"])}
    elif kind == "tones":
        out = {"sig": tones(case["seed"], case["B"], case["T"])}
    elif kind == "decode":
        return {"toks": torch.from_numpy(prng.randint(case["seed"], "toks", (case["B"], case["N"], case["K"]), 2048))}
    else:
        raise ValueError(kind)
    if "length" in case:
        out["length"] = torch.tensor(case["length"], dtype=torch.float32)
    return out
