"""Round 5: a captured encode replayed on NEW data against the eager call (Mimi tiny)."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from golden_cases import noise
from audiocodecs_amd import Mimi, checkpoint
from audiocodecs_amd.config import MIMI_TINY
sd = checkpoint.synthetic_mimi_state_dict(MIMI_TINY, seed=0)
eager = Mimi(24000, num_codebooks=4, state_dict=sd, config=MIMI_TINY).eval()
cap = Mimi(24000, num_codebooks=4, state_dict=sd, config=MIMI_TINY).eval()
a, b = noise(1, 1, 9600).cuda(), (noise(2, 1, 9600) * 3).cuda()
with torch.no_grad():
    cap.sig_to_toks(a); torch.cuda.synchronize()
    sx = a.clone()
    g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            so = cap.sig_to_feats(sx)
    for name, x in (("A", a), ("B (3x louder)", b), ("A again", a), ("B again", b)):
        sx.copy_(x); g.replay(); torch.cuda.synchronize()
        ref = eager.sig_to_feats(x)
        print(f"replay on {name}: max |graph - eager| = {float((so - ref).abs().max()):.3e}, equal = {bool(torch.equal(so, ref))}")
