"""Round 4: every launch of one EnCodec encode + decode call at batch 1 (the reference's own regime), per-launch events with shapes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AC_PROF_DETAIL"] = "1"
import bench
from audiocodecs_amd import prng
codec, cfg, sd = bench.build_codec("encodec")
from audiocodecs_amd._native import debug_set
TAP8 = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("tap8=")]
sys.argv = [a for a in sys.argv if not a.startswith("tap8=")]
for B, sec in [(int(a.split("x")[0]), float(a.split("x")[1])) for a in sys.argv[1:]] or [(1, 1.0), (1, 10.0)]:
    T = int(round(sec * cfg.sampling_rate))
    sig = torch.from_numpy((prng.normal(123, "bench.sig.lat", (B, T)) * 0.1).astype(np.float32)).cuda()
    with torch.no_grad():
        codec.sig_to_toks(sig)
        if TAP8: debug_set(codec, "tap8", TAP8[0])
        for _ in range(3): codec.toks_to_sig(codec.sig_to_toks(sig))
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(20): codec.toks_to_sig(codec.sig_to_toks(sig))
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20
        st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(5)])
    tot = sum(s[2] for s in st) / 5
    print(f"\n{B} clip(s) x {sec} s: {wall * 1e3:.3f} ms per call (host), kernel events {tot:.3f} ms, {sum(s[1] for s in st) / 5:.0f} launches")
    for s in sorted(st, key=lambda s: -s[2])[:16]:
        print(f"   {s[2] / 5:7.3f} ms  x{s[1] / 5:<3.0f} {s[0]}")
