"""Child process of tests/test_bench_flow_gloo.py: one rank of bench.run() over gloo with a stub codec on CPU tensors.
(No GPU call anywhere: torch.distributed is initialised first, the stub stands in for the HIP library.)"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class StubCodec:
    """sig_to_toks / toks_to_sig / profile_kernels with the wrappers' shapes (hop 320, 8 codebooks) and the library's kernel names."""

    def __init__(self, rank):
        self.rank, self.calls = rank, 0

    def sig_to_toks(self, sig):
        self.calls += 1
        n = -(-sig.shape[1] // 320)
        return (torch.arange(sig.shape[0] * n * 8).reshape(sig.shape[0], n, 8) + self.rank) % 1024

    def toks_to_sig(self, toks):
        return torch.zeros(toks.shape[0], toks.shape[1] * 320)

    def profile_kernels(self, fn):
        fn()
        return [("tap_gemm6_kernel<1, 4, 4, 1, 2>", 70, 3.0, 9.0e12, 8.0e9), ("tap_gemm6_kernel<1, 4, 4, 2, 2, dil>", 50, 2.9, 8.6e12, 8.4e9),
                ("lstm_persist16_kernel<true>", 20, 4.1, 8.0e11, 1.4e9), ("rvq_decode_kernel", 10, 0.02, 5e7, 2.2e8)]


def main():
    import bench

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "2", "--seconds", "0.5", "--no-cpu-baseline"])
    from audiocodecs_amd.config import ENCODEC_24KHZ as cfg

    sig = torch.zeros(2, 12000)
    codec = StubCodec(rank)
    rc = bench.run(args, codec, cfg, None, sig, sig, rank, world, dist, torch.device("cpu"))
    assert codec.calls == 2 * 3 + 1, codec.calls      # warm-up + the profiled K steps + the plain K steps, nothing rank-conditional
    dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
