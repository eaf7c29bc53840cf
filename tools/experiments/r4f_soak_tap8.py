"""Round 4 race screen for tap_gemm8: many encode+decode steps of Mimi (linear layers in row mode, one tap: the two-register-set
pipeline) and EnCodec with tap_gemm8 FORCED wherever the shape allows, every step's tokens and waveform compared bit for bit with a
tap_gemm6-only run of the same handle (a weight fragment still in flight across a stage barrier would show as rare wrong tiles)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from audiocodecs_amd import prng
from audiocodecs_amd._native import debug_set

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, B in (("mimi", 128), ("encodec", 64), ("dac", 16)):
    codec, cfg, sd = bench.build_codec(name)
    sig = torch.from_numpy((prng.normal(99, "soak8." + name, (B, int(10 * cfg.sampling_rate))) * 0.1).astype(np.float32)).cuda()
    with torch.no_grad():
        codec.sig_to_toks(sig[:1])
        debug_set(codec, "tap8", 0)
        t0 = codec.sig_to_toks(sig); r0 = codec.toks_to_sig(t0); torch.cuda.synchronize()
        bad = 0
        tic = time.time()
        for form in (0, 1, 2, 3):
            debug_set(codec, "tap8", 1); debug_set(codec, "tap8_form", form)
            for i in range(n // 4):
                t = codec.sig_to_toks(sig); r = codec.toks_to_sig(t)
                bad += int(not torch.equal(t, t0)) + int(not torch.equal(r, r0))
        torch.cuda.synchronize()
    print(name, "steps", n // 4 * 4, "with tap_gemm8 forced (forms model/1/2/3), checks that differ from the tap_gemm6 run:", bad, f"{(time.time()-tic)/(n//4*4)*1e3:.2f} ms/step", flush=True)
    assert bad == 0
    bench.drop_codec(codec)
