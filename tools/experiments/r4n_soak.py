"""Round 4: rerun screen of the final library -- every codec, a throughput-sized and a batch-1-sized call (the batch-1 regime runs
kernels the big batches never do: tiny launches on tap_gemm8, the shared-group codebook search), N repetitions each, every output
compared bit for bit with the first.  A kernel whose result depends on timing (a missed hazard, a race on LDS or on the in-order
memory counter) shows up here as a differing repetition."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from audiocodecs_amd import prng
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, shapes in (("encodec", ((64, 10.0), (16, 4.0), (1, 1.0), (3, 0.417))), ("mimi", ((8, 4.0), (1, 1.0))), ("wavtokenizer", ((8, 3.0), (1, 1.0))), ("dac", ((4, 2.0), (1, 0.5)))):
    codec, cfg, sd = bench.build_codec(name)
    for B, sec in shapes:
        T = int(round(sec * cfg.sampling_rate))
        sig = torch.from_numpy((prng.normal(321, f"soak.{name}", (B, T)) * 0.1).astype(np.float32)).cuda()
        t0 = time.time(); bad = 0
        with torch.no_grad():
            toks0 = codec.sig_to_toks(sig); rec0 = codec.toks_to_sig(toks0); feats0 = codec.sig_to_feats(sig)
            for i in range(N):
                toks = codec.sig_to_toks(sig); rec = codec.toks_to_sig(toks0); feats = codec.sig_to_feats(sig)
                if not (torch.equal(toks, toks0) and torch.equal(rec, rec0) and torch.equal(feats, feats0)):
                    bad += 1
                    if bad <= 3: print(f"  {name} B={B} {sec}s repetition {i}: tokens {bool(torch.equal(toks, toks0))} waveform {bool(torch.equal(rec, rec0))} feats {bool(torch.equal(feats, feats0))}", flush=True)
        print(f"{name:13s} B={B:2d} x {sec:5.3f} s: {N} repetitions, {bad} differ from the first ({time.time() - t0:.0f} s)", flush=True)
