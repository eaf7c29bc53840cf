"""Phase / stage stamps of one Mimi tap_gemm6 layer (a -DT6_TRACE build; see r3o_trace.py).  AC_TRACE_SHAPE=0,N,K picks the layer."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from golden_cases import noise
from audiocodecs_amd import Mimi, checkpoint
from audiocodecs_amd.config import MIMI_24KHZ as cfg
sd = checkpoint.synthetic_mimi_state_dict(cfg, seed=0)
codec = Mimi(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
sig = noise(777, 128, 240000).cuda()
toks = codec.sig_to_toks(sig)
nat = next(iter(codec._natives.values())); L = nat.lib
mhz = C.c_double(0)
L.ac_debug_clock(nat.h, 1, C.byref(mhz))
codec.toks_to_sig(codec.sig_to_toks(sig))
buf = (C.c_ulonglong * (16 + 8 * 16 * 8))()
n = L.ac_debug_trace(nat.h, buf, len(buf))
ph = np.array(buf[4:16], dtype=np.int64)
lab = {1: "tile index", 2: "scales / slots", 3: "enter_segment", 4: "A + B loads issued", 5: "A arrived, split, stored", 6: "barrier (loop starts)", 7: "main loop done",
       8: "epi: col tile 0 bias arrived", 10: "col tile 1", 11: "all stores issued", 9: "end"}
print("wave 1 phases (cycles since entry): " + ", ".join(f"{lab[k]} +{ph[k]-ph[0]}" for k in (1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 9) if ph[k]))
a = np.array(buf[16:n], dtype=np.int64).reshape(8, 16, 8)
for w in (1,):
    st = a[w]
    for s in range(16):
        if st[s, 0] <= 0 or st[s, 5] == 0: continue
        d = [int(st[s, 0] - st[s, 6])] + [int(st[s, k] - st[s, k - 1]) for k in range(1, 6)]
        print(f"   stage {s:2d}: " + " ".join(f"{x:6d}" for x in d) + f" | {int(st[s,5]-st[s,0]):6d} flags interior={st[s,7]&1} new_chunk={(st[s,7]>>1)&1} mt={st[s,7]>>8}")
L.ac_debug_clock(nat.h, 0, C.byref(mhz)); print("shader MHz", mhz.value)
