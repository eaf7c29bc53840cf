"""Round 6: per-layer times of one encode + decode step (prof_detail: one record per tap-GEMM shape), EnCodec 64 x 10 s."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set
name = os.environ.get("CODEC", "encodec")
batch = {"mimi": 128, "encodec": 64, "wavtokenizer": 64, "dac": 39}[name]
codec, cfg, sd = bench.build_codec(name)
T = int(round(10.0 * cfg.sampling_rate))
sig = torch.from_numpy((prng.normal(123, f"bench.sig.{name}", (batch, T)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    codec.toks_to_sig(codec.sig_to_toks(sig)); torch.cuda.synchronize()
    debug_set(codec, "prof_detail", 1)
    st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(5)])
tot = 0.0
for s in st:
    ms = s[2] / 5; tot += ms
    if ms > 0.02:
        print(f"{ms:8.3f} ms  x{s[1] // 5:<3d} {s[3] / s[2] / 1e9 / 1e0 if s[2] else 0:8.1f} TF/s-eq {s[4] / s[2] / 1e6 if s[2] else 0:8.1f} GB/s  {s[0]}")
print(f"total {tot:.3f} ms")
