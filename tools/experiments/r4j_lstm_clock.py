"""Round 4: the shader clock during lstm_persist16 (trace, lstm_dbg bit 5) -- with the input projection fused or not, and with silent clips."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from audiocodecs_amd import Encodec, checkpoint
from audiocodecs_amd._native import debug_set
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg
from golden_cases import noise
sd = checkpoint.synthetic_state_dict(cfg, seed=0)
c = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
c.sig_to_feats(noise(5, 1, 24000).cuda())
for fuse in (1, 0):
    for B, silent in ((16, 0), (16, 1), (1, 0), (64, 0)):
        sig = noise(5, B, 240000).cuda()
        if silent: sig[1:] = 0
        debug_set(c, "lstm_fuse_in", fuse); debug_set(c, "lstm_dbg", 0)
        c.sig_to_feats(sig)
        debug_set(c, "lstm_dbg", 32)
        print(f"fuse_in={fuse} B={B} silent_but_first={silent}", file=sys.stderr, flush=True)
        c.sig_to_feats(sig); torch.cuda.synchronize()
debug_set(c, "lstm_dbg", 0)
