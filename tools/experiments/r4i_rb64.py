"""Round 4: rb_fused6<64> / rb128 / enc_front / dec_tail launch times of the EnCodec step (64 x 10 s) for the library named by AUDIOCODECS_AMD_LIB."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
codec, cfg, sd = bench.build_codec("encodec")
from golden_cases import noise
sig = noise(5, 64, 240000).cuda()
with torch.no_grad():
    for _ in range(2): toks = codec.sig_to_toks(sig); rec = codec.toks_to_sig(toks)
    st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(5)])
tot = 0.0
for s in st:
    tot += s[2] / 5
    if any(k in s[0] for k in ("rb_fused6", "rb128", "enc_front", "dec_tail", "lstm")): print(f"{s[0]:40s} {s[2]/5:.3f} ms per step", flush=True)
print(f"sum of kernel events {tot:.2f} ms per step; tokens {int(toks.sum())} waveform {float(rec.double().abs().sum()):.6f}")
