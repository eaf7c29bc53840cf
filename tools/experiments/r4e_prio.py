"""Round 4: s_setprio experiments on tap_gemm8 (tap_stagger bits: 1 = static priority for waves 4..7, 2 = priority 1 around every unit's MFMAs)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set
name = sys.argv[1] if len(sys.argv) > 1 else "encodec"
batch = {"mimi": 128, "wavtokenizer": 64, "dac": 32, "encodec": 64}[name]
codec, cfg, sd = bench.build_codec(name)
sig = torch.from_numpy((prng.normal(123, f"bench.sig.{name}", (batch, int(10 * cfg.sampling_rate))) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    codec.toks_to_sig(codec.sig_to_toks(sig))
    debug_set(codec, "tap8", -1)
    for rep in range(3):
        for st in (0, 1, 2, 3):
            debug_set(codec, "tap_stagger", st)
            codec.toks_to_sig(codec.sig_to_toks(sig)); torch.cuda.synchronize()
            rows = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(5)])
            t8 = sum(r[2] for r in rows if r[0].startswith("tap_gemm8")) / 5
            print(f"{name} rep {rep} prio bits {st}: tap_gemm8 launches {t8:.3f} ms per step, step {sum(r[2] for r in rows) / 5:.3f} ms", flush=True)
