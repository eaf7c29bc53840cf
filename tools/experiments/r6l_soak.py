"""Round 6 soak of the stream kernels: EnCodec (64 x 10 s) and Mimi (32 x 10 s) encode + decode repeated; EVERY step's tokens and waveform must be
bit-identical to the first step's (a race or a hazard shows as a run-to-run difference: profiles/r3_pk_fma_hazard.md), the sticky status clean."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from audiocodecs_amd import Encodec, Mimi, checkpoint, prng
from audiocodecs_amd.config import ENCODEC_24KHZ, MIMI_24KHZ

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for name, B in (("encodec", 64), ("mimi", 32)):
    if name == "encodec":
        codec = Encodec(24000, num_codebooks=8, state_dict=checkpoint.synthetic_state_dict(ENCODEC_24KHZ, seed=0)).eval()
    else:
        codec = Mimi(24000, num_codebooks=8, state_dict=checkpoint.synthetic_mimi_state_dict(MIMI_24KHZ, seed=0), config=MIMI_24KHZ).eval()
    sig = torch.from_numpy((prng.normal(99, "soak", (B, 240000)) * 0.1).astype(np.float32)).cuda()
    with torch.no_grad():
        t0 = codec.sig_to_toks(sig); r0 = codec.toks_to_sig(t0); torch.cuda.synchronize()
        bad = 0
        tic = time.time()
        for i in range(n):
            t = codec.sig_to_toks(sig); r = codec.toks_to_sig(t)
            bad += int(not torch.equal(t, t0)) + int(not torch.equal(r, r0))
        torch.cuda.synchronize()
    print(name, "steps", n, "mismatching checks", bad, f"{(time.time() - tic) / n * 1e3:.2f} ms/step (with the comparisons)", flush=True)
    assert bad == 0
