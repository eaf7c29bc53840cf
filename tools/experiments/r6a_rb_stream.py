"""Round 6: rb_stream6 (16 waves per CU, weights in LDS, one wave = one stream) against rb_fused6<64> in ONE process:
outputs compared bit for bit, per-kernel times from the library's own events.  Usage: python tools/experiments/r6a_rb_stream.py [encodec|mimi]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set
name = sys.argv[1] if len(sys.argv) > 1 else "encodec"
batch = {"mimi": 128, "encodec": 64}[name]
codec, cfg, sd = bench.build_codec(name)
T = int(round(10.0 * cfg.sampling_rate))
sig = torch.from_numpy((prng.normal(123, f"bench.sig.{name}", (batch, T)) * 0.1).astype(np.float32)).cuda()
res = {}
with torch.no_grad():
    codec.sig_to_toks(sig[:2])
    for mode in [int(m) for m in os.environ.get('RB_MODES', '1,0,1,0').split(',')]:
        debug_set(codec, "rb_stream", mode)
        toks = codec.sig_to_toks(sig); wav = codec.toks_to_sig(toks); torch.cuda.synchronize()
        st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(5)])
        rb = {}
        for s in st:
            k = s[0].split("(")[0]
            if "rb_" in k: rb[k] = rb.get(k, 0.0) + s[2] / 5
        tot = sum(s[2] for s in st) / 5
        print(f"rb_stream={mode}: step kernels {tot:.3f} ms | " + " ".join(f"{k}={v:.3f}" for k, v in sorted(rb.items())), flush=True)
        res[mode] = (toks.cpu().numpy(), wav.cpu().numpy())
ks = sorted(res); t1, w1 = res[ks[-1]]; t0, w0 = res[ks[0]]
print("tokens differing:", int((t1 != t0).sum()), "of", t1.size)
d = (w1.astype(np.float64) - w0.astype(np.float64))
print("waveform: bit-equal" if np.array_equal(w1, w0) else f"waveform max abs diff {np.abs(d).max():.3e} rms {np.sqrt((d**2).mean()):.3e} (signal rms {np.sqrt((w0.astype(np.float64)**2).mean()):.3e})")
