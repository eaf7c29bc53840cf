"""Case list for the DAC golden fixtures (shared by tools/make_golden_dac.py and the tests).

The fixtures come from the STAND-IN (transformers.DacModel, same architecture) because the reference's
backend (descript-audio-codec) is not installed here: parity with the reference itself is unpinned
(oracle/dac_oracle.py header).  Inputs are re-drawn from the repo PRNG on both sides."""

from __future__ import annotations

import torch

from audiocodecs_amd import prng
from golden_cases import noise, tones

REC_STRIDE = 61

CASES = [
    # 44.1 kHz architecture (BASELINE.json configs[2]): strides (2,4,8,8), hop 512, 9 codebooks
    dict(name="full_noise_b2", cfg="full", weights_seed=0, kind="noise", B=2, T=22050, seed=211, K=9),
    dict(name="full_tones_K8", cfg="full", weights_seed=0, kind="tones", B=1, T=16000, seed=212, K=8),
    dict(name="full_T1023", cfg="full", weights_seed=0, kind="noise", B=1, T=1023, seed=213, K=9),  # 1 frame; T < 512 raises upstream
    dict(name="full_T512", cfg="full", weights_seed=0, kind="noise", B=2, T=512, seed=214, K=9),
    dict(name="full_T513", cfg="full", weights_seed=0, kind="noise", B=1, T=513, seed=215, K=9),
    dict(name="full_T4097_K1", cfg="full", weights_seed=0, kind="noise", B=1, T=4097, seed=216, K=1),
    dict(name="full_decode_rand", cfg="full", weights_seed=0, kind="decode", B=2, N=9, K=9, seed=221),
    dict(name="full_decode_K3", cfg="full", weights_seed=0, kind="decode", B=1, N=4, K=3, seed=222),
    dict(name="full_w1_noise", cfg="full", weights_seed=1, kind="noise", B=1, T=6000, seed=231, K=9),
    # tiny architecture (1/8 width, strides (2,4,5,8) incl. an odd one): every module output stored
    dict(name="tiny_taps", cfg="tiny", weights_seed=0, kind="noise", B=2, T=6400, seed=241, K=4, taps=True),
    dict(name="tiny_odd", cfg="tiny", weights_seed=0, kind="noise", B=3, T=3333, seed=242, K=3, taps=True),
]


def make_input(case: dict, golden_dir: str) -> dict:
    kind = case["kind"]
    if kind == "noise":
        return {"sig": noise(case["seed"], case["B"], case["T"])}
    if kind == "tones":
        return {"sig": tones(case["seed"], case["B"], case["T"])}
    if kind == "decode":
        return {"toks": torch.from_numpy(prng.randint(case["seed"], "toks", (case["B"], case["N"], case["K"]), 1024))}
    raise ValueError(kind)
