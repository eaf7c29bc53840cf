"""Round 4 debugging aid: direct vs staged tap-GEMM epilogue (ac_debug_set "tap_epi_staged") on small EnCodec batches, per tap8 setting:
which module outputs differ first."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from audiocodecs_amd._native import debug_set
from golden_cases import noise
codec, cfg, sd = bench.build_codec("encodec")
B, T = 3, 9999
sig = noise(6160 + T, B, T).cuda()
with torch.no_grad():
    codec.sig_to_toks(sig[:1])
    for tap8 in (0, -1, 1):
        debug_set(codec, "tap8", tap8)
        res = []
        for staged in (0, 1, 0, 1):
            debug_set(codec, "tap_epi_staged", staged)
            f = codec.sig_to_feats(sig); t = codec.sig_to_toks(sig)
            res.append((f.clone(), t.clone()))
        debug_set(codec, "tap_epi_staged", 0)
        print(f"tap8={tap8}: feats direct==staged {torch.equal(res[0][0], res[1][0])}  (direct rerun equal {torch.equal(res[0][0], res[2][0])}, staged rerun equal {torch.equal(res[1][0], res[3][0])});"
              f" toks direct==staged {torch.equal(res[0][1], res[1][1])} (reruns {torch.equal(res[0][1], res[2][1])}, {torch.equal(res[1][1], res[3][1])});"
              f" max |df| {float((res[0][0] - res[1][0]).abs().max()):.3e}")
with torch.no_grad():
    debug_set(codec, "tap8", -1)
    for rv in (1, 0):
        debug_set(codec, "rvq_exact", rv)
        ts = [codec.sig_to_toks(sig).clone() for _ in range(6)]
        print(f"rvq_exact={rv}: reruns equal {[bool(torch.equal(ts[0], t)) for t in ts[1:]]}; differing entries {[int((ts[0] != t).sum()) for t in ts[1:]]}")
        if not rv:
            d = (ts[0] != ts[1]).nonzero()
            print("first differences (clip, frame, stage):", d[:6].tolist(), ts[0][ts[0] != ts[1]][:6].tolist(), ts[1][ts[0] != ts[1]][:6].tolist())
