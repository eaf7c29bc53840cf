"""Run-to-run bit equality of the encoder features / decoded waveform at full size, per library build and fusion switch."""
import os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from golden_cases import noise
from audiocodecs_amd import Encodec, checkpoint
from audiocodecs_amd.config import ENCODEC_24KHZ as cfg
sd = checkpoint.synthetic_state_dict(cfg, seed=0)
codec = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
sig = noise(777, 64, 240000).cuda()
f0 = codec.sig_to_feats(sig); t0 = codec.sig_to_toks(sig); r0 = codec.toks_to_sig(t0)
bad_f = bad_r = 0
for rep in range(6):
    f = codec.sig_to_feats(sig); r = codec.toks_to_sig(t0)
    bad_f += int((f != f0).any(dim=(1, 2)).sum()); bad_r += int((r != r0).any(dim=1).sum())
print(os.environ.get("AUDIOCODECS_AMD_LIB", "default lib"), "AC_FUSE", os.environ.get("AC_FUSE"), ": clips differing across 6 reruns: feats", bad_f, "waveform", bad_r)
