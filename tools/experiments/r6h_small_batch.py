"""Round 6: do the thin stages run faster per clip when the whole inter-layer tensor fits the 256 MB memory-side cache?  Per-kernel times of one
encode + decode at B = 2, 4, 8, 16, 64 clips x 10 s, scaled to 64 clips."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
codec, cfg, sd = bench.build_codec("encodec")
T = 240000
full = torch.from_numpy((prng.normal(123, "bench.sig.encodec", (64, T)) * 0.1).astype(np.float32)).cuda()
pats = ("enc_stream", "rb_stream6", "dec_stream", "rb128", "tap_gemm")
with torch.no_grad():
    for B in (64, 16, 8, 4, 2, 64):
        sig = full[:B].contiguous()
        for _ in range(2): codec.toks_to_sig(codec.sig_to_toks(sig))
        torch.cuda.synchronize()
        st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(5)])
        r = {}
        for s in st:
            for p in pats:
                if p in s[0]: r[p] = r.get(p, 0.0) + s[2] / 5
        print(f"B {B:3d}: " + " ".join(f"{k}={v * 64 / B:.3f}" for k, v in r.items()) + "   (ms per step, scaled to 64 clips)", flush=True)
