"""Round 5: per-kernel times of ONE library build (AUDIOCODECS_AMD_LIB), every kernel whose name holds one of the substrings given:
   AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_new.so python tools/experiments/r5u_kernel_ab.py new mimi rb_fused6 rb128"""
import os, sys, time, hashlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
tag, name, pats = sys.argv[1], sys.argv[2], sys.argv[3:]
batch = {"mimi": 128, "wavtokenizer": 64, "dac": 39, "encodec": 64}[name]
codec, cfg, sd = bench.build_codec(name)
T = int(round(10.0 * cfg.sampling_rate))
sig = torch.from_numpy((prng.normal(123, f"bench.sig.{name}", (batch, T)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    for _ in range(3): codec.toks_to_sig(codec.sig_to_toks(sig))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): codec.toks_to_sig(codec.sig_to_toks(sig))
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 10 * 1e3
    st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(5)])
    _t = codec.sig_to_toks(sig); _r = codec.toks_to_sig(_t); torch.cuda.synchronize()
digest = hashlib.sha256(_t.cpu().numpy().tobytes() + _r.cpu().numpy().tobytes()).hexdigest()[:12]
r = {}
for s in st:
    k = s[0].split("(")[0]
    if not pats or any(p in k for p in pats):
        r[k] = r.get(k, 0.0) + s[2] / 5
print(f"{tag} {name} [{digest}]: step {step:.3f} ms | " + " ".join(f"{k.replace('ac::', '').replace('void ', '')}={v:.3f}" for k, v in sorted(r.items())), flush=True)
