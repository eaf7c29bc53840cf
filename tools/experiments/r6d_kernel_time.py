"""Round 6: times of the kernels whose name holds one of the given substrings, for the library AUDIOCODECS_AMD_LIB points at (encoder +
decoder pass of EnCodec 64 x 10 s, five repeats).  Usage: AUDIOCODECS_AMD_LIB=... python tools/experiments/r6d_kernel_time.py tag pat..."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from audiocodecs_amd import prng
tag, pats = sys.argv[1], sys.argv[2:]
name = os.environ.get("CODEC", "encodec")
batch = {"mimi": 128, "encodec": 64, "wavtokenizer": 64, "dac": 39}[name]
codec, cfg, sd = bench.build_codec(name)
T = int(round(10.0 * cfg.sampling_rate))
sig = torch.from_numpy((prng.normal(123, f"bench.sig.{name}", (batch, T)) * 0.1).astype(np.float32)).cuda()
with torch.no_grad():
    codec.sig_to_toks(sig[:1])
    if os.environ.get('CHAIN'):
        from audiocodecs_amd._native import debug_set
        debug_set(codec, 'chain_stream', int(os.environ['CHAIN']))
    if os.environ.get('RB128'):
        from audiocodecs_amd._native import debug_set
        debug_set(codec, 'rb128_stream', int(os.environ['RB128']))
    if os.environ.get('RB_STREAM'):
        from audiocodecs_amd._native import debug_set
        debug_set(codec, 'rb_stream', int(os.environ['RB_STREAM']))
    toks = codec.sig_to_toks(sig); codec.toks_to_sig(toks); torch.cuda.synchronize()
    st = codec.profile_kernels(lambda: [codec.toks_to_sig(codec.sig_to_toks(sig)) for _ in range(5)])
r = {}
for s in st:
    k = s[0].split("(")[0]
    if any(p in k for p in pats): r[k] = r.get(k, 0.0) + s[2] / 5
print(f"{tag}: step {sum(s[2] for s in st) / 5:.3f} ms | " + " ".join(f"{k}={v:.3f}" for k, v in sorted(r.items())), flush=True)
