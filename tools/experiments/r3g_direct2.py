import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from audiocodecs_amd import Encodec, checkpoint, _native
from audiocodecs_amd.config import ENCODEC_24KHZ
import test_workspace_contract_gpu as M
from golden_cases import noise
sd = checkpoint.synthetic_state_dict(ENCODEC_24KHZ, seed=0)
codec = Encodec(24000, num_codebooks=8, state_dict=sd, config=ENCODEC_24KHZ).eval()
codec.sig_to_toks(noise(3, 1, 640).cuda())
enc = (codec, next(iter(codec._natives.values())))
M.test_calls_with_a_growing_batch_are_graph_capturable(enc)
print("direct call ok", flush=True)
