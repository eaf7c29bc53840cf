"""Round 6: is the per-launch cost model of the tap-GEMMs leaving anything on the table?  Per-layer times of one EnCodec step with the tile
arrangement forced (ac_debug_set "tap8" / "tap8_form" / "tap_pick"), one process."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set
codec, cfg, sd = bench.build_codec("encodec")
sig = torch.from_numpy((prng.normal(123, "bench.sig.encodec", (64, 240000)) * 0.1).astype(np.float32)).cuda()
def run(tag, **sw):
    for k, v in sw.items(): debug_set(codec, k, v)
    with torch.no_grad():
        codec.toks_to_sig(codec.sig_to_toks(sig)); torch.cuda.synchronize()
        st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(3)])
    rows = {}
    for s in st:
        if "tap_gemm" in s[0]:
            shape = s[0].split("> ")[-1] if "> " in s[0] else s[0]
            rows[shape] = (s[2] / 3, s[0].split(" B64")[0])
    return rows
with torch.no_grad():
    codec.sig_to_toks(sig[:1])
debug_set(codec, "prof_detail", 1)
base = run("model")
alts = {"tap8=0": dict(tap8=0), "tap8=1": dict(tap8=1), "tap8=1 form1": dict(tap8=1, tap8_form=1), "tap8=1 form2": dict(tap8=1, tap8_form=2), "tap8=1 form3": dict(tap8=1, tap8_form=3),
        "pick0": dict(tap8=0, tap8_form=0, tap_pick=0), "pick1": dict(tap8=0, tap_pick=1), "pick2": dict(tap8=0, tap_pick=2)}
res = {k: run(k, **v) for k, v in alts.items()}
print(f"{'layer':42s} {'model':>8s} " + " ".join(f"{k:>13s}" for k in alts))
tot = 0.0; best = 0.0
for shape, (ms, name) in base.items():
    vals = [res[k].get(shape, (float('nan'), ''))[0] for k in alts]
    tot += ms; best += min([ms] + [v for v in vals if v == v])
    print(f"{shape[:42]:42s} {ms:8.3f} " + " ".join(f"{v:13.3f}" for v in vals) + f"   {name.replace('tap_gemm', 'tg')}")
print(f"sum model {tot:.3f} ms, sum of per-layer minima {best:.3f} ms")
