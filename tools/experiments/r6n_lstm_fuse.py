import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set
codec, cfg, sd = bench.build_codec("encodec")
sig = torch.from_numpy((prng.normal(123, "bench.sig.encodec", (64, 240000)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    codec.sig_to_toks(sig[:1])
    for v in (1, 0, 1, 0):
        debug_set(codec, "lstm_fuse_in", v)
        codec.toks_to_sig(codec.sig_to_toks(sig)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): codec.toks_to_sig(codec.sig_to_toks(sig))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(3)])
        l = sum(s[2] for s in st if "lstm" in s[0]) / 3
        g = sum(s[2] for s in st if "tap_gemm" in s[0]) / 3
        print(f"lstm_fuse_in={v}: step {ms:.3f} ms, lstm kernels {l:.3f}, tap_gemm {g:.3f}", flush=True)
