import os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from golden_cases import noise
from test_gpu_parity import capture
from audiocodecs_amd import Encodec, checkpoint
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg
sd = checkpoint.synthetic_state_dict(cfg, seed=0)
codec = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
T = 9600
sig = noise(4141, 41, T).cuda()
def taps(x):
    B = x.shape[0]
    f, flat = capture(codec, lambda: codec.sig_to_feats(x), 1 << 27)
    n0 = B * T * 32
    x0 = flat[:n0].reshape(B, T, 32); y1 = flat[n0:2*n0].reshape(B, T, 32); y2 = flat[2*n0:2*n0 + B*(T//2)*64].reshape(B, T//2, 64)
    return x0, y1, y2
codec.sig_to_feats(sig[:1])
os.environ["AC_FRONT_SEG"] = "1000"
ref = taps(sig[:20])
for seg, pad in (("1", "0"), ("1", "4096"), ("2", "0"), ("2", "4096"), ("1", "0")):
    os.environ["AC_FRONT_SEG"] = seg; os.environ["AC_FRONT_LDSPAD"] = pad
    got = taps(sig[:20])
    for name, a, b in zip(("x0", "y1", "y2"), ref, got):
        d = np.abs(a - b).max(axis=(2,))
        bad = np.argwhere(d > 0)
        print("seg", seg, "pad", pad, name, "rows differing:", len(bad), bad[:12].tolist(), "max", float(d.max()))
# decoder tail: segmentation independence at many streams
os.environ.pop("AC_FRONT_SEG", None); os.environ.pop("AC_FRONT_LDSPAD", None)
g = torch.Generator().manual_seed(5)
toks = torch.randint(0, 1024, (40, 60, 8), generator=g).cuda()
os.environ["AC_TAIL_SEG"] = "1000"
r0 = codec.toks_to_sig(toks)
for seg in ("1", "2", "5"):
    os.environ["AC_TAIL_SEG"] = seg
    for rep in range(2):
        r = codec.toks_to_sig(toks)
        print("tail seg", seg, "rep", rep, "equal:", bool(torch.equal(r, r0)), "ndiff", int((r != r0).sum()))
