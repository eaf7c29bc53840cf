"""Round 5: phase stamps of one tap_gemm8 workgroup on Mimi's transformer linear layers (row mode, one tap; -DT6_TRACE build):
   AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_trace.so python tools/experiments/r5r_trace_mimi.py"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench
from audiocodecs_amd import prng
codec, cfg, sd = bench.build_codec("mimi")
sig = torch.from_numpy((prng.normal(123, "bench.sig.mimi", (128, 240000)) * 0.1).astype(np.float32)).cuda()
toks = codec.sig_to_toks(sig)
nat = next(iter(codec._natives.values())); L = nat.lib
mhz = C.c_double(0)
L.ac_debug_clock(nat.h, 1, C.byref(mhz))
for name, shp in (("fc1 (GELU)", "32000,2048,512"), ("fc2 (LayerScale + residual)", "32000,512,2048"), ("qkv", "32000,1536,512"), ("o_proj", "32000,512,512")):
    os.environ["AC_TRACE_SHAPE"] = shp
    buf = (C.c_ulonglong * (16 + 8 * 16 * 8))()
    L.ac_debug_trace(nat.h, buf, len(buf))
    codec.sig_to_feats(sig); torch.cuda.synchronize()
    n = L.ac_debug_trace(nat.h, buf, len(buf))
    ph = np.array(buf[4:16], dtype=np.int64)
    lab = {1: "tile index", 2: "scales", 6: "first barrier (loop starts)", 7: "main loop done", 8: "epilogue pass 0 staged", 10: "pass 0 stored", 11: "all stores issued", 9: "end"}
    print(f"== {name} (M,N,K = {shp}): wave 1 phases (ticks since entry): " + ", ".join(f"{lab[k]} +{ph[k]-ph[0]}" for k in (1, 2, 6, 7, 8, 10, 11, 9) if ph[k]))
    a = np.array(buf[16:n], dtype=np.int64).reshape(8, 16, 8)
    for w in (0, 4):
        st = a[w]; ok = (st[:, 0] > 0) & (st[:, 4] > 0)
        if ok.any():
            d = st[ok, 4] - st[ok, 0]
            print(f"   wave {w}: {int(ok.sum())} stages, mean stage {d.mean():.0f} ticks")
L.ac_debug_clock(nat.h, 0, C.byref(mhz)); print("shader MHz", mhz.value)
