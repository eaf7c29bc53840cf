import sys, torch, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from golden_cases import noise
from audiocodecs_amd import Encodec, checkpoint, _native
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg
mode = sys.argv[1]
sd = checkpoint.synthetic_state_dict(cfg, seed=0)
codec = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
T, K, Bmax = 4800, 8, 40
sig = noise(11, Bmax, T).cuda()
import os
if os.environ.get("FIRST640"): codec.sig_to_toks(noise(3, 1, 640).cuda())
codec.sig_to_toks(sig[:2])
nat = next(iter(codec._natives.values())); L = nat.lib
N = codec.config.num_frames(T)
ws_bytes = max(L.ac_encode_workspace_bytes(nat.h, Bmax, T), L.ac_decode_workspace_bytes(nat.h, Bmax, N))
ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
sizes = {"one": (7,), "two": (2, 7), "four": (2, 7, 40, 3), "sep": (2, 7, 40, 3), "big": (40,), "enc2": (2, 7), "dec2": (2, 7)}[mode]
toks = {B: torch.empty(B, N, K, dtype=torch.int64, device="cuda") for B in sizes}
rec = {B: torch.empty(B, N * 320, device="cuda") for B in sizes}
if mode == "dec2":
    for B in sizes: toks[B].copy_(codec.sig_to_toks(sig[:B]))
torch.cuda.synchronize()
side = torch.cuda.Stream()
graphs = []
with torch.cuda.stream(side):
    groups = [[B] for B in sizes] if mode == "sep" else [list(sizes)]
    for grp in groups:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for B in grp:
                if mode != "dec2": _native.check(L.ac_encode(nat.h, P(sig), None, B, T, K, P(toks[B]), P(ws), ws_bytes, S()), nat.h, "enc")
                if mode != "enc2": _native.check(L.ac_decode(nat.h, P(toks[B]), B, N, K, P(rec[B]), P(ws), ws_bytes, S()), nat.h, "dec")
        graphs.append(g)
print(mode, "captured", flush=True)
if os.environ.get("ZERO"):
    for B in sizes: toks[B].zero_(); rec[B].zero_()
for g in graphs:
    g.replay()
torch.cuda.synchronize()
print(mode, "replayed", flush=True)
ok = all(torch.equal(toks[B], codec.sig_to_toks(sig[:B])) for B in sizes)
print(mode, "tokens equal", ok)
