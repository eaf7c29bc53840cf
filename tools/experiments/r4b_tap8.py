"""Round 4: tap_gemm8 (csrc/tap_gemm8.h) against tap_gemm6 on the EnCodec step, 64 x 10 s: (1) bit-equality of features, tokens and
waveform between the two kernels (same arithmetic in the same order), (2) per-layer times (AC_PROF_DETAIL=1) in ONE process
(ac_debug_set flips the kernel between timed passes: same box, same clocks)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AC_PROF_DETAIL"] = "1"
from audiocodecs_amd import Encodec, checkpoint, prng
from audiocodecs_amd._native import debug_set
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg

sd = checkpoint.synthetic_state_dict(cfg, seed=0)
codec = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
modes = [tuple(int(x) for x in (a.split(":") + ["0"])[:2]) for a in sys.argv[2:]] or [(0, 0), (1, 0)]      # tap8[:form]
sig = torch.from_numpy((prng.normal(123, "bench.sig.rank0", (B, 240000)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    ref = None
    for mode in modes:
        codec.sig_to_toks(sig[:1])
        debug_set(codec, "tap8", mode[0]); debug_set(codec, "tap8_form", mode[1])
        feats = codec.sig_to_feats(sig)
        toks = codec.sig_to_toks(sig)
        rec = codec.toks_to_sig(toks)
        torch.cuda.synchronize()
        if ref is None:
            ref = (feats, toks, rec)
        else:
            print(f"tap8={mode} vs tap8={modes[0]}: feats equal {torch.equal(feats, ref[0])} (max abs diff {float((feats - ref[0]).abs().max()):.3e}), "
                  f"tokens equal {torch.equal(toks, ref[1])} ({int((toks != ref[1]).sum())} differ), waveform equal {torch.equal(rec, ref[2])} "
                  f"(max abs diff {float((rec - ref[2]).abs().max()):.3e})", flush=True)
    for rep in range(2):
        for mode in modes:
            debug_set(codec, "tap8", mode[0]); debug_set(codec, "tap8_form", mode[1])
            for _ in range(2):
                codec.toks_to_sig(codec.sig_to_toks(sig))
            torch.cuda.synchronize()
            st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(10)])
            rows = [(s[0], s[1] / 10, s[2] / 10, s[3] / (s[2] * 1e-3) / 1e12 if s[2] else 0) for s in st]
            tap = [r for r in rows if r[0].startswith("tap_gemm")]
            print(f"== pass {rep}, tap8={mode}: tap-GEMM launches {sum(r[2] for r in tap):.3f} ms per step; whole step (event sum) {sum(r[2] for r in rows):.3f} ms", flush=True)
            if rep == 1:
                for r in sorted(tap, key=lambda r: -r[2]):
                    print(f"   {r[2]:.3f} ms  x{r[1]:.0f}  {r[3]:6.1f} TF  {r[0]}")
