"""Stage stamps of one tap_gemm6 workgroup (a -DT6_TRACE build): where do the cycles of a K stage go?
   AC_OUT=tools/experiments/lib_trace.so AC_OBJ=/tmp/obj_trace bash audiocodecs_amd/csrc/build.sh -DT6_TRACE
   AUDIOCODECS_AMD_LIB=$PWD/tools/experiments/lib_trace.so python tools/experiments/r3o_trace.py"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from golden_cases import noise
from audiocodecs_amd import Encodec, checkpoint
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg
sd = checkpoint.synthetic_state_dict(cfg, seed=0)
codec = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
sig = noise(777, 64, 240000).cuda()
codec.sig_to_feats(sig)
nat = next(iter(codec._natives.values())); L = nat.lib
mhz = C.c_double(0)
L.ac_debug_clock(nat.h, 1, C.byref(mhz))
# every tap_gemm6 launch of the encoder overwrites the stamps: the LAST launch with >= 16 stages wins -> run the encoder, read
codec.toks_to_sig(codec.sig_to_toks(sig))
buf = (C.c_ulonglong * (16 + 8 * 16 * 8))()
n = L.ac_debug_trace(nat.h, buf, len(buf))
ph = np.array(buf[4:16], dtype=np.int64)
lab = {1: "tile index", 2: "scales / slots", 3: "enter_segment", 4: "A + B loads issued", 5: "A arrived, split, stored", 6: "barrier (loop starts)", 7: "main loop done",
       8: "epilogue: bias/winv of column tile 0 arrived (+ everything older)", 10: "column tile 0 stored and drained, tile 1's bias/winv arrived", 11: "all stores issued", 9: "amax flushed (end)"}
print("wave 1 phases (cycles since entry): " + ", ".join(f"{lab[k]} +{ph[k]-ph[0]}" for k in (1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 9) if ph[k]))
a = np.array(buf[16:n], dtype=np.int64).reshape(8, 16, 8)
names = ["start", "entry->A loads issued", "k0 issued", "k1 issued", "A arrived", "staged", "barrier"]
for w in (0, 1, 2, 3):
    st = a[w]
    ok = st[:, 0] > 0
    if not ok.any(): continue
    print(f"wave {w}: per stage deltas (cycles): " + ", ".join(names[1:]) + " | stage total")
    for s in range(16):
        if not ok[s] or st[s, 5] == 0: continue
        d = [int(st[s, 0] - st[s, 6])] + [int(st[s, k] - st[s, k - 1]) for k in range(1, 6)]
        nxt = int(st[s + 1, 0] - st[s, 0]) if s + 1 < 16 and ok[s + 1] else -1
        print(f"   stage {s:2d}: " + " ".join(f"{x:6d}" for x in d) + f" | {int(st[s,5]-st[s,0]):6d}  (next stage starts +{nxt}) flags interior={st[s,7]&1} new_chunk={(st[s,7]>>1)&1} mt={st[s,7]>>8}")
L.ac_debug_clock(nat.h, 0, C.byref(mhz)); print("shader MHz", mhz.value)
