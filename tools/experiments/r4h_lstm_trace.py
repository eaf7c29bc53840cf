"""Round 4: lstm_persist16's in-kernel stamps (lstm_dbg bit 5) by number of clips.  EnCodec encoder, 10 s clips."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from audiocodecs_amd import Encodec, checkpoint
from audiocodecs_amd._native import debug_set
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg
from golden_cases import noise
sd = checkpoint.synthetic_state_dict(cfg, seed=0)
c = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
c.sig_to_feats(noise(5, 1, 24000).cuda())
for B in [int(a) for a in sys.argv[1:]] or (1, 16, 64):
    sig = noise(5, B, 240000).cuda()
    debug_set(c, "lstm_dbg", 0)
    c.sig_to_feats(sig)
    debug_set(c, "lstm_dbg", 32)
    print("B =", B, file=sys.stderr, flush=True)
    c.sig_to_feats(sig); torch.cuda.synchronize()
