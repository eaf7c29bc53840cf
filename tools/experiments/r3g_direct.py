import sys, os, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from audiocodecs_amd import Encodec, checkpoint, _native
from audiocodecs_amd.config import ENCODEC_24KHZ
from golden_cases import noise
sizes = tuple(int(x) for x in sys.argv[1].split(","))
order = sys.argv[2] if len(sys.argv) > 2 else "ws_first"
_ptr = lambda t: C.c_void_p(t.data_ptr())
_stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
sd = checkpoint.synthetic_state_dict(ENCODEC_24KHZ, seed=0)
codec = Encodec(24000, num_codebooks=8, state_dict=sd, config=ENCODEC_24KHZ).eval()
codec.sig_to_toks(noise(3, 1, 640).cuda())
nat = next(iter(codec._natives.values())); L = nat.lib
T, K, Bmax = 4800, 8, 40
N = codec.config.num_frames(T)
ws_bytes = max(L.ac_encode_workspace_bytes(nat.h, Bmax, T), L.ac_decode_workspace_bytes(nat.h, Bmax, N))
if order == "ws_first":
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    sig = noise(11, Bmax, T).cuda()
    want_t = codec.sig_to_toks(sig[:2])
else:
    sig = noise(11, Bmax, T).cuda()
    want_t = codec.sig_to_toks(sig[:2])
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
print("ws", hex(ws.data_ptr()), ws_bytes, "natws", hex(nat.ws.data_ptr()), nat.ws.numel(), flush=True)
toks = {B: torch.empty(B, N, K, dtype=torch.int64, device="cuda") for B in sizes}
rec = {B: torch.empty(B, N * 320, device="cuda") for B in sizes}
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for B in sizes:
            _native.check(L.ac_encode(nat.h, _ptr(sig), None, B, T, K, _ptr(toks[B]), _ptr(ws), ws_bytes, _stream()), nat.h, "ac_encode")
            _native.check(L.ac_decode(nat.h, _ptr(toks[B]), B, N, K, _ptr(rec[B]), _ptr(ws), ws_bytes, _stream()), nat.h, "ac_decode")
g.replay()
torch.cuda.synchronize()
print(sizes, order, "replay ok", all(torch.equal(toks[B], codec.sig_to_toks(sig[:B])) for B in sizes), flush=True)
