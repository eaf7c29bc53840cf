"""Stage stamps of one tap_gemm8 workgroup (a -DT6_TRACE build): where do the cycles of a tile go?
   AC_OUT=tools/experiments/lib_trace.so AC_OBJ=/tmp/obj_trace bash audiocodecs_amd/csrc/build.sh -DT6_TRACE
   AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_trace.so python tools/experiments/r5a_trace8.py
Per traced layer (AC_TRACE_SHAPE = "M,N,K" picks it inside the library): phases of wave 1 (entry -> scales -> first barrier -> main loop done
-> epilogue marks) and, per wave, the mean / max over the first 16 stages of [issue (weights + activation requests), compute (fragment
reads + MFMAs + slab writes), counted wait, barrier]."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from golden_cases import noise
from audiocodecs_amd import Encodec, checkpoint
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg
sd = checkpoint.synthetic_state_dict(cfg, seed=0)
codec = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
sig = noise(777, 64, 240000).cuda()
toks = codec.sig_to_toks(sig)
nat = next(iter(codec._natives.values())); L = nat.lib
mhz = C.c_double(0)
L.ac_debug_clock(nat.h, 1, C.byref(mhz))
SHAPES = [("down x5", "6000,256,1280", "enc"), ("down x8", "750,512,4096", "enc"), ("down x4", "30000,128,512", "enc"), ("rb256 k3", "6000,128,768", "enc"),
          ("up x4", "30000,256,256", "dec"), ("up x8", "750,2048,1024", "dec"), ("up x5", "6000,640,512", "dec")]
for name, shp, side in SHAPES:
    os.environ["AC_TRACE_SHAPE"] = shp
    buf = (C.c_ulonglong * (16 + 8 * 16 * 8))()
    L.ac_debug_trace(nat.h, buf, len(buf))          # clear
    if side == "enc": codec.sig_to_feats(sig)
    else: codec.toks_to_sig(toks)
    torch.cuda.synchronize()
    n = L.ac_debug_trace(nat.h, buf, len(buf))
    ph = np.array(buf[4:16], dtype=np.int64)
    print(f"== {name} (M,N,K = {shp})")
    lab = {1: "tile index", 2: "scales", 6: "first barrier (loop starts)", 7: "main loop done", 8: "epi: col tile 0 constants", 10: "epi: col tile 1", 11: "all stores issued", 9: "end"}
    print("   wave 1 phases (cycles since entry): " + ", ".join(f"{lab[k]} +{ph[k]-ph[0]}" for k in (1, 2, 6, 7, 8, 10, 11, 9) if ph[k]))
    a = np.array(buf[16:n], dtype=np.int64).reshape(8, 16, 8)
    for w in range(8):
        st = a[w]
        ok = (st[:, 0] > 0) & (st[:, 4] > 0)
        if not ok.any(): continue
        d = np.stack([st[ok, 1] - st[ok, 0], st[ok, 2] - st[ok, 1], st[ok, 3] - st[ok, 2], st[ok, 4] - st[ok, 3], st[ok, 4] - st[ok, 0]], 1)
        print(f"   wave {w}: {int(ok.sum())} stages; mean issue {d[:,0].mean():.0f} compute {d[:,1].mean():.0f} wait {d[:,2].mean():.0f} barrier {d[:,3].mean():.0f} | stage {d[:,4].mean():.0f} (max {d[:,4].max()}, min {d[:,4].min()})")
    for sidx, word in ((2, 5), (3, 6)):      # unit stamps of stages 2 and 3: deltas between the units of each wave
        rows = []
        for w in range(8):
            us = a[w, 8:16, word]
            if us[0] and a[w, sidx, 1]:
                t = np.concatenate([[a[w, sidx, 1]], us])
                rows.append(f"w{w}: " + " ".join(f"{int(x):4d}" for x in np.diff(t)))
        if rows: print(f"   stage {sidx} units (cycles from the stage's start / previous unit, per wave): " + " | ".join(rows))
    w = 1
    st = a[w]
    for s_ in range(16):
        if st[s_, 0] and st[s_, 4]:
            print(f"      wave 1 stage {s_:2d}: issue {st[s_,1]-st[s_,0]:5d} compute {st[s_,2]-st[s_,1]:5d} wait {st[s_,3]-st[s_,2]:5d} barrier {st[s_,4]-st[s_,3]:5d}")
L.ac_debug_clock(nat.h, 0, C.byref(mhz)); print("shader MHz", mhz.value)
