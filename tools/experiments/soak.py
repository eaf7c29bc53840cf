"""Soak: many encode+decode steps; every output must stay bit-identical to the first step's and the handle's sticky status clean."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from audiocodecs_amd import Encodec, WavTokenizer, checkpoint, prng
from audiocodecs_amd.config import ENCODEC_24KHZ, WAVTOK_40

for name in ("encodec", "wavtokenizer"):
    if name == "encodec":
        codec = Encodec(24000, num_codebooks=8, state_dict=checkpoint.synthetic_state_dict(ENCODEC_24KHZ, seed=0)).eval()
    else:
        codec = WavTokenizer(24000, state_dict=checkpoint.synthetic_wavtok_state_dict(WAVTOK_40, seed=0), arch=WAVTOK_40).eval()
    sig = torch.from_numpy((prng.normal(99, "soak", (64, 240000)) * 0.1).astype(np.float32)).cuda()
    t0 = codec.sig_to_toks(sig); r0 = codec.toks_to_sig(t0); torch.cuda.synchronize()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    bad = 0
    tic = time.time()
    for i in range(n):
        t = codec.sig_to_toks(sig); r = codec.toks_to_sig(t)
        if i % 25 == 24:
            bad += int(not torch.equal(t, t0)) + int(not torch.equal(r, r0))
    torch.cuda.synchronize()
    nat = next(iter(codec._natives.values()))
    print(name, "steps", n, "mismatching checks", bad, "lstm_status", nat.lib.ac_lstm_status(nat.h), f"{(time.time()-tic)/n*1e3:.2f} ms/step", flush=True)
    assert bad == 0 and nat.lib.ac_lstm_status(nat.h) == 1
