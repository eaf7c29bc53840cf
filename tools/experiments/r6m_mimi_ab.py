"""Round 6: Mimi 128 x 10 s, one process, alternating passes with a developer switch on / off: whole-step time (host clock around K steps) and
the kernels' event times by family.  Usage: python tools/experiments/r6m_mimi_ab.py <switch> [passes]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set
key = sys.argv[1]; passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
codec, cfg, sd = bench.build_codec("mimi")
sig = torch.from_numpy((prng.normal(123, "bench.sig.mimi", (128, 240000)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    codec.toks_to_sig(codec.sig_to_toks(sig)); torch.cuda.synchronize()
    for i in range(2 * passes):
        v = 1 - (i & 1)
        debug_set(codec, key, v)
        codec.toks_to_sig(codec.sig_to_toks(sig)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): codec.toks_to_sig(codec.sig_to_toks(sig))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(3)])
        fam = {}
        for s in st:
            k = s[0].split("<")[0].split(" ")[0]
            fam[k] = fam.get(k, 0.0) + s[2] / 3
        print(f"{key}={v}: step {ms:.2f} ms | " + " ".join(f"{k}={x:.2f}" for k, x in sorted(fam.items(), key=lambda kv: -kv[1])[:7]), flush=True)
