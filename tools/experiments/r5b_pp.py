"""Round 5: per-layer tap-GEMM times of one codec at its BASELINE.json per-GPU size -- tap_gemm8's ping-pong main loop (tap8_spread = 1) against the
lock-step one and against tap_gemm6, in ONE process (ac_debug_set between timed passes).  Usage: r5b_pp.py <codec> [batch] [tap8:form:pp ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AC_PROF_DETAIL"] = "1"
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set

name = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else {"mimi": 128, "wavtokenizer": 64, "dac": 39, "encodec": 64}[name]
modes = [tuple(int(x) for x in (a.split(":") + ["0", "1", "0"])[:4]) for a in sys.argv[3:]] or [(-1, 0, 0, 0), (-1, 0, 1, 0), (-1, 0, 2, 0)]
codec, cfg, sd = bench.build_codec(name)
T = int(round(10.0 * cfg.sampling_rate))
sig = torch.from_numpy((prng.normal(123, f"bench.sig.{name}", (batch, T)) * 0.1).astype(np.float32)).cuda()
res, ref = {}, None
with torch.no_grad():
    codec.toks_to_sig(codec.sig_to_toks(sig))
    for rep in range(2):
        for mode in modes:
            debug_set(codec, "tap8", mode[0]); debug_set(codec, "tap8_form", mode[1]); debug_set(codec, "tap8_spread", mode[2])
            toks = codec.sig_to_toks(sig); rec = codec.toks_to_sig(toks)
            torch.cuda.synchronize()
            if ref is None: ref = (toks.clone(), rec.clone())
            same = bool(torch.equal(toks, ref[0]) and torch.equal(rec, ref[1]))
            st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(3)])
            res[mode] = {}
            for s in st:
                if s[0].startswith("tap_gemm"):
                    shape = s[0].split("> ", 1)[1] if "> " in s[0] else s[0]
                    k = res[mode].setdefault(shape, [0.0, 0, "", 0.0])
                    k[0] += s[2] / 3; k[1] += s[1] / 3; k[2] = s[0].split(" B")[0]; k[3] += s[3] / 3
            print(f"pass {rep} (tap8, form, spread, wreg)={mode}: tap-GEMM {sum(v[0] for v in res[mode].values()):.3f} ms, step (event sum) {sum(s[2] for s in st) / 3:.3f} ms, outputs {'EQUAL' if same else 'DIFFER'}", flush=True)
base = modes[0]
print(f"\nper layer shape (ms per step; launches), {modes}")
for shape, v in sorted(res[base].items(), key=lambda kv: -kv[1][0]):
    row = f"{shape:42s} x{v[1]:<4.0f}"
    for m in modes:
        w = res[m].get(shape)
        row += f" | {w[0]:7.3f} {w[3] / (w[0] * 1e-3) / 1e12:6.1f} TF {w[2].replace('tap_gemm', 'g').replace('_kernel', '')}" if w else " | -"
    print(row)
