"""Round 5: per-layer tap-GEMM times of ONE library build (AUDIOCODECS_AMD_LIB) -- run once per build, alternating, on the same box:
   for i in 1 2 3; do for l in old new; do AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_$l.so python tools/experiments/r5l_lib_ab.py $l; done; done"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AC_PROF_DETAIL"] = "1"
import bench
from audiocodecs_amd import prng
tag = sys.argv[1] if len(sys.argv) > 1 else "?"
name = sys.argv[2] if len(sys.argv) > 2 else "encodec"
batch = {"mimi": 128, "wavtokenizer": 64, "dac": 39, "encodec": 64}[name]
codec, cfg, sd = bench.build_codec(name)
T = int(round(10.0 * cfg.sampling_rate))
sig = torch.from_numpy((prng.normal(123, f"bench.sig.{name}", (batch, T)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    for _ in range(3): codec.toks_to_sig(codec.sig_to_toks(sig))
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(10): codec.toks_to_sig(codec.sig_to_toks(sig))
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 10 * 1e3
    st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(5)])
r = {}
for s in st:
    if s[0].startswith("tap_gemm"):
        shape = s[0].split("> ", 1)[1] if "> " in s[0] else s[0]
        r[shape] = r.get(shape, 0.0) + s[2] / 5
import hashlib
with torch.no_grad():
    _t = codec.sig_to_toks(sig); _r = codec.toks_to_sig(_t); torch.cuda.synchronize()
digest = hashlib.sha256(_t.cpu().numpy().tobytes() + _r.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"{tag} {name} [{digest}]: step {step:.3f} ms, tap-GEMM {sum(r.values()):.3f} ms | " + " ".join(f"{v:.3f}" for k, v in sorted(r.items())), flush=True)
