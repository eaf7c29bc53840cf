"""Round 4: ms per step of the four codecs at their BASELINE.json per-GPU sizes under developer switches given as KEY=VALUE arguments
(one process per setting; switches are latched at ac_finalize).  Usage: r4c_codecs.py [codec,codec,...] -- AC_TAP8=0 -- AC_TAP8=1 ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json, time, numpy as np, torch
sys.path.insert(0, %r)
import bench
from audiocodecs_amd import prng
name, batch = sys.argv[1], int(sys.argv[2])
codec, cfg, sd = bench.build_codec(name)
T = int(round(10.0 * cfg.sampling_rate))
sig = torch.from_numpy((prng.normal(123, f"bench.sig.{name}", (batch, T)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    toks = codec.sig_to_toks(sig); rec = codec.toks_to_sig(toks)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): codec.toks_to_sig(codec.sig_to_toks(sig))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    st = codec.profile_kernels(lambda: codec.toks_to_sig(codec.sig_to_toks(sig)))
    gate = bench.parity_gate(name, codec)
fam = {}
for s in st:
    k = s[0].split("<")[0]
    fam[k] = fam.get(k, 0.0) + s[2]
print(json.dumps({"ms": dt * 1e3, "toks_sum": int(toks.sum()), "rec_sum": float(rec.double().abs().sum()), "fam": sorted(fam.items(), key=lambda kv: -kv[1])[:5],
                  "parity": [gate["token_exact_match"], gate["decode_rms_err"]]}))
''' % ROOT
args = sys.argv[1:]
codecs = ["mimi", "wavtokenizer", "dac"]
if args and "=" not in args[0] and args[0] != "--":
    codecs = args[0].split(","); args = args[1:]
settings, cur = [], {}
for a in args:
    if a == "--":
        if cur: settings.append(cur)
        cur = {}
    else:
        k, v = a.split("=", 1); cur[k] = v
if cur: settings.append(cur)
settings = settings or [{}]
B = {"mimi": 128, "wavtokenizer": 64, "dac": 64, "encodec": 64}
for name in codecs:
    base = None
    for st in settings:
        out = subprocess.run([sys.executable, "-c", CHILD, name, str(B[name])], env=dict(os.environ, **st), capture_output=True, text=True)
        if out.returncode:
            print(name, st, "FAILED", out.stderr[-800:]); continue
        r = json.loads(out.stdout.strip().splitlines()[-1])
        same = "" if base is None else f"  tokens {'EQUAL' if r['toks_sum'] == base['toks_sum'] else 'DIFFER'}, waveform {'EQUAL' if r['rec_sum'] == base['rec_sum'] else 'DIFFERS'} vs first"
        base = base or r
        print(f"{name:13s} {st}: {r['ms']:.2f} ms per step  parity {r['parity']}{same}\n      " + ", ".join(f"{k} {v:.2f}" for k, v in r["fam"]), flush=True)
