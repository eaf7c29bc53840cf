"""N > 1 path on CPU: world_size-2 gloo processes shard clips and gather tokens."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audiocodecs_amd.sharding import gather_tokens, shard_bounds


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_clips, q, vocab=1024):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = (torch.arange(num_clips * 5 * 8).reshape(num_clips, 5, 8) * 7) % vocab
    lo, hi = shard_bounds(num_clips, rank, world)
    got = gather_tokens(full[lo:hi].clone(), num_clips)
    q.put((rank, bool(torch.equal(got, full)), got.dtype == torch.int64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_clips", [8, 7, 1])
def test_gather_tokens_two_ranks(num_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and dt for _, ok, dt in res), res


def test_gather_tokens_mimi_vocabulary():
    """Mimi ids reach 2047: still exact through the int16 wire format."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, q, 2048)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and dt for _, ok, dt in res), res


def test_shard_bounds_cover_batch_once():
    for n in (1, 7, 64, 65):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
