"""Round 6: codebook search with tile t's argmax beside tile t + 1's MFMAs against the library AUDIOCODECS_AMD_LIB_BASE points at: tokens
must be identical.  Usage: python tools/experiments/r6r_rvq_pipe.py (runs itself once per library)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for tag, lib in (("base", os.environ.get("AUDIOCODECS_AMD_LIB_BASE", os.path.join(ROOT, "libac_base.so"))), ("new", "")) * 2:
        env = dict(os.environ)
        if lib:
            env["AUDIOCODECS_AMD_LIB"] = lib
        subprocess.run([sys.executable, __file__, tag], env=env, check=True)
    import torch
    a, b = torch.load("/tmp/r6r_base.pt"), torch.load("/tmp/r6r_new.pt")
    print("tokens identical:", [bool(torch.equal(x, y)) for x, y in zip(a, b)])
    sys.exit(0)
import numpy as np, torch
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
codec, cfg, sd = bench.build_codec("encodec")
outs = []
for B, T in [(64, 240000), (45, 237777)]:
    sig = torch.from_numpy((prng.normal(11, f"r6r.{B}.{T}", (B, T)) * 0.1).astype(np.float32)).cuda()
    with torch.no_grad():
        toks = codec.sig_to_toks(sig)
        st = codec.profile_kernels(lambda: [codec.sig_to_toks(sig) for _ in range(5)])
    outs.append(toks.cpu())
    print(f"{sys.argv[1]}: B={B} T={T} rvq_encode {sum(s[2] for s in st if 'rvq_encode' in s[0]) / 5:.4f} ms", flush=True)
torch.save(outs, f"/tmp/r6r_{sys.argv[1]}.pt")
