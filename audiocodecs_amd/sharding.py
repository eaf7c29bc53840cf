"""Clip sharding across the GPUs of one node.

Clips are independent units (no cross-clip state anywhere on the path), so rank r simply owns the
contiguous slice [r*ceil(B/G), ...) of the batch; weights are replicated.  The ONLY collective is
the gather of the token ids after encode -- one all_gather over RCCL/xGMI (backend "nccl" on ROCm),
a few hundred KB per GPU.  Decode needs none.  The reference itself never shards the codec (it is
replicated per SpeechBrain DDP rank, downstream/test_sr.py:350-351).
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "shard_clips", "gather_tokens"]


def shard_bounds(num_clips: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) owned by `rank`; the last shards may be short or empty."""
    per = -(-num_clips // world)
    lo = min(rank * per, num_clips)
    return lo, min(lo + per, num_clips)


def shard_clips(batch: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(batch.shape[0], rank, world)
    return batch[lo:hi]


def gather_tokens(toks: torch.Tensor, num_clips: Optional[int] = None, group=None, force: bool = False) -> torch.Tensor:
    """toks [b_local, N, K] int64 on every rank -> [num_clips, N, K] on every rank, in clip order.

    Token ids fit 16 bits for every configured codec (codebook <= 4096 < 32768), so they travel as
    int16 (4x fewer bytes on the wire) when they do; shards are padded to the common size
    ceil(num_clips/world) so a single fixed-size all_gather suffices for ragged last shards.
    `force=True` runs the collective even at world size 1 (exercises the RCCL path on a single GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        return toks
    if dist.get_world_size(group) == 1 and not force:
        return toks
    world = dist.get_world_size(group)
    n_local = toks.shape[0]
    if num_clips is None:
        num_clips = n_local * world
    per = -(-num_clips // world)
    wire = toks.to(torch.int16)
    if n_local < per:
        pad = torch.zeros((per - n_local,) + tuple(toks.shape[1:]), dtype=wire.dtype, device=toks.device)
        wire = torch.cat([wire, pad])
    out = torch.empty((world * per,) + tuple(toks.shape[1:]), dtype=wire.dtype, device=toks.device)
    # neither RCCL nor gloo has an int16 type; a gather moves bytes, so ship the raw bytes
    dist.all_gather_into_tensor(out.view(torch.uint8), wire.contiguous().view(torch.uint8), group=group)
    return out[:num_clips].to(torch.int64)
