"""Round 5: does de-phasing the CUs' first tiles (random start delays up to `tap_stagger` cycles) help the tap-GEMMs whose tiles run in lock step
across the chip (few tiles per CU: all epilogues -- HBM write bursts -- at the same time)?  Usage: r5g_stagger.py [stagger values ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AC_PROF_DETAIL"] = "1"
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set
vals = [int(a) for a in sys.argv[1:]] or [0, 20000, 60000, 150000, 0]
codec, cfg, sd = bench.build_codec("encodec")
sig = torch.from_numpy((prng.normal(123, "bench.sig.encodec", (64, 240000)) * 0.1).astype(np.float32)).cuda()
res = {}
with torch.no_grad():
    codec.toks_to_sig(codec.sig_to_toks(sig))
    for i, v in enumerate(vals):
        debug_set(codec, "tap_stagger", v)
        codec.toks_to_sig(codec.sig_to_toks(sig)); torch.cuda.synchronize()
        st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(3)])
        r = {}
        for s in st:
            if s[0].startswith("tap_gemm"):
                shape = s[0].split("> ", 1)[1] if "> " in s[0] else s[0]
                r[shape] = r.get(shape, 0.0) + s[2] / 3
        res[(i, v)] = r
        print(f"stagger {v}: tap-GEMM {sum(r.values()):.3f} ms", flush=True)
keys = list(res)
for shape in sorted(res[keys[0]], key=lambda k: -res[keys[0]][k]):
    print(f"{shape:42s} " + " | ".join(f"{res[k][shape]:.3f}" for k in keys))
