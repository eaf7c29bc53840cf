"""Case list for the WavTokenizer fixtures (shared by tools/make_golden_wavtok.py, the tests and bench.py's parity gate).

PARITY UNPINNED: the reference's backend package (`wavtokenizer`, lucadellalib/WavTokenizer) is not on disk, so these
fixtures are outputs of oracle/wavtokenizer_oracle.py (the restatement of its published modules), NOT of the reference.
What they pin is the HIP path to the oracle, and the oracle to itself across rounds.  The encoder + codebook search
part of the oracle IS cross-checked against an independent third-party implementation of the same published SEANet
(transformers' EncodecModel with use_causal_conv=False): tests/test_wavtok_oracle_golden.py.
hop = 6*5*5*4 = 600 samples; vocabulary 4096; one codebook."""

from __future__ import annotations

import torch

from audiocodecs_amd import prng
from golden_cases import noise, read_example_wav, tones

REC_STRIDE = 61

CASES = [
    # 40 tok/s architecture (wavtokenizer_smalldata_frame40_...yaml), weights seed 0
    dict(name="full_example", cfg="full", weights_seed=0, kind="wav"),
    dict(name="full_noise_b2", cfg="full", weights_seed=0, kind="noise", B=2, T=24000, seed=311),
    dict(name="full_T1", cfg="full", weights_seed=0, kind="noise", B=2, T=1, seed=313),
    dict(name="full_T3", cfg="full", weights_seed=0, kind="noise", B=1, T=3, seed=319),
    dict(name="full_T599", cfg="full", weights_seed=0, kind="noise", B=1, T=599, seed=314),
    dict(name="full_T600", cfg="full", weights_seed=0, kind="noise", B=2, T=600, seed=315),
    dict(name="full_T601", cfg="full", weights_seed=0, kind="noise", B=1, T=601, seed=316),
    dict(name="full_T2477", cfg="full", weights_seed=0, kind="noise", B=3, T=2477, seed=317),
    dict(name="full_tones_b2", cfg="full", weights_seed=0, kind="tones", B=2, T=36000, seed=318),
    dict(name="full_decode_rand", cfg="full", weights_seed=0, kind="decode", B=2, N=70, seed=321),
    dict(name="full_decode_N1", cfg="full", weights_seed=0, kind="decode", B=1, N=1, seed=322),
    dict(name="full_w1_noise", cfg="full", weights_seed=1, kind="noise", B=1, T=9000, seed=331),
    # 75 tok/s architecture (hop 320, n_fft 1280)
    dict(name="f75_noise", cfg="f75", weights_seed=0, kind="noise", B=2, T=8000, seed=351),
    # tiny architecture (hop 48): every module output is in the fixture
    dict(name="tiny_taps", cfg="tiny", weights_seed=0, kind="noise", B=2, T=4800, seed=341, taps=True),
    dict(name="tiny_odd", cfg="tiny", weights_seed=0, kind="noise", B=3, T=1111, seed=342, taps=True),
]


def make_input(case: dict, golden_dir: str) -> dict:
    kind = case["kind"]
    if kind == "wav":
        return {"sig": read_example_wav(golden_dir)}
    if kind == "noise":
        return {"sig": noise(case["seed"], case["B"], case["T"])}
    if kind == "tones":
        return {"sig": tones(case["seed"], case["B"], case["T"])}
    if kind == "decode":
        vocab = 128 if case["cfg"] == "tiny" else 4096
        return {"toks": torch.from_numpy(prng.randint(case["seed"], "toks", (case["B"], case["N"], 1), vocab))}
    raise ValueError(kind)
