"""RCCL on the device (world size 1 -- the GPU box has one MI355X): the token gather of sharding.py runs through
`all_gather_into_tensor` on the "nccl" (= RCCL) backend with the int16-on-the-wire byte views, at config-2 size.
The 1 -> 8 GPU curve itself is the driver's to measure (bench.py --gpus N); this pins the collective's plumbing on HBM."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gather_tokens_through_rccl_world_size_one():
    from audiocodecs_amd.sharding import gather_tokens, shard_clips

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        toks = torch.randint(0, 1024, (64, 750, 8), device="cuda")          # config 2: 64 clips x 750 frames x 8 codebooks
        got = gather_tokens(toks, 64, force=True)
        torch.cuda.synchronize()
        assert got.is_cuda and got.dtype == torch.int64 and torch.equal(got, toks)
        big = torch.randint(0, 4096, (64, 400, 1), device="cuda")           # WavTokenizer ids reach 4095: still exact as int16
        assert torch.equal(gather_tokens(big, 64, force=True), big)
        assert shard_clips(toks).shape[0] == 64
        dist.barrier()
    finally:
        dist.destroy_process_group()
