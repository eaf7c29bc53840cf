"""Round 5: which phase of rb_fused6<64> costs what -- the kernel's timing modes (rb6_dbg; results are WRONG in every mode but 0):
1 no stage-A MFMAs (and their fragment reads), 2 no stage-B MFMAs, 4 no staging of the next tile, 8 no output epilogue, 16 no loads."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set
codec, cfg, sd = bench.build_codec("encodec")
sig = torch.from_numpy((prng.normal(123, "bench.sig.encodec", (64, 240000)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    codec.toks_to_sig(codec.sig_to_toks(sig))
    for mode in (0, 1, 2, 3, 4, 8, 12, 16, 7, 15, 31, 0):
        debug_set(codec, "rb6_dbg", mode)
        codec.sig_to_feats(sig); torch.cuda.synchronize()
        st = codec.profile_kernels(lambda: [codec.sig_to_feats(sig) for _ in range(3)])
        rb = [s for s in st if s[0].startswith("rb_fused6_kernel<64")]
        print(f"rb6_dbg {mode:2d}: rb_fused6<64> {sum(s[2] for s in rb) / 3:.3f} ms per encoder pass ({sum(s[1] for s in rb) // 3} launch)", flush=True)
    debug_set(codec, "rb6_dbg", 0)
