import sys, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from audiocodecs_amd import Encodec, checkpoint
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg
from golden_cases import noise
sd = checkpoint.synthetic_state_dict(cfg, seed=0)
c = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
sig = noise(5, 64, 240000).cuda()
for _ in range(2): c.sig_to_feats(sig)
st = c.profile_kernels(lambda: [c.sig_to_feats(sig) for _ in range(5)])
for s in st:
    if "lstm" in s[0]: print(s[0], round(s[2]/5,3), "ms per LSTM ->", round(s[2]/5/750*1e3,2), "us/step")
