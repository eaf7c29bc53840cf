"""Round 4: per-layer tap-GEMM times of the EnCodec step (64 x 10 s) by tile arrangement -- the cost model's picks against the
256-row, 8-wave arrangements forced on every layer that can take them (AC_TAP_PICK=3: 256 x 256, 4: 256 x 128).  One process per
arrangement (the switches are latched per handle at ac_finalize); prints one line per (arrangement, layer)."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, json, numpy as np, torch
sys.path.insert(0, %r)
from audiocodecs_amd import Encodec, checkpoint, prng
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg
sd = checkpoint.synthetic_state_dict(cfg, seed=0)
codec = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
sig = torch.from_numpy((prng.normal(123, "bench.sig.rank0", (64, 240000)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    for _ in range(3): t = codec.sig_to_toks(sig); codec.toks_to_sig(t)
    torch.cuda.synchronize()
    st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(10)])
print(json.dumps([(s[0], s[1] / 10, s[2] / 10, s[3] / (s[2] * 1e-3) / 1e12 if s[2] else 0) for s in st]))
''' % ROOT

res = {}
for pick in ("", "3", "4"):
    env = dict(os.environ, AC_PROF_DETAIL="1")
    if pick:
        env["AC_TAP_PICK"] = pick
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    if out.returncode:
        print("pick", pick or "model", "FAILED", out.stderr[-1500:])
        continue
    res[pick or "model"] = json.loads(out.stdout.strip().splitlines()[-1])
for pick, rows in res.items():
    tot = sum(r[2] for r in rows if r[0].startswith("tap_gemm6"))
    print(f"== AC_TAP_PICK={pick}: tap-GEMM launches {tot:.3f} ms per step; whole step (event sum) {sum(r[2] for r in rows):.3f} ms")
    for r in sorted((r for r in rows if r[0].startswith("tap_gemm6")), key=lambda r: -r[2]):
        print(f"   {r[2]:.3f} ms  x{r[1]:.0f}  {r[3]:6.1f} TF  {r[0]}")
