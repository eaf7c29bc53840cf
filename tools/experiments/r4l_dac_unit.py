"""Round 4: DAC 32 x 10 s with the 64- / 96-channel residual units fused (dac_unit6_kernel) or as two launches: per-kernel-name times."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AC_PROF_DETAIL"] = "1"
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set
codec, cfg, sd = bench.build_codec("dac")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(round(10.0 * cfg.sampling_rate))
sig = torch.from_numpy((prng.normal(123, "bench.sig.dac", (B, T)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    codec.toks_to_sig(codec.sig_to_toks(sig))
    for unit in (0, 1, 0, 1):
        debug_set(codec, "dac_unit", unit)
        codec.toks_to_sig(codec.sig_to_toks(sig)); torch.cuda.synchronize()
        st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(2)])
        tot = sum(s[2] for s in st) / 2
        print(f"\ndac_unit={unit}: step (event sum) {tot:.2f} ms")
        rows = [s for s in st if ("N64 " in s[0] or "N96 " in s[0] or "dac_unit" in s[0])]
        for s in sorted(rows, key=lambda s: -s[2]): print(f"   {s[2] / 2:8.3f} ms x{s[1] / 2:<3.0f} {s[0]}")
        print(f"   sum of the 64- / 96-channel layers: {sum(s[2] for s in rows) / 2:.2f} ms")
