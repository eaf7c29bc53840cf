"""Repo-owned counter-based PRNG (numpy only, exactly reproducible everywhere).

Every tensor is addressed by ``(seed, name)``; element ``i`` of that tensor depends only on
``(seed, name, i)``.  Only integer arithmetic and exactly-rounded IEEE-754 double operations
(+, -, *) are used -- no libm transcendental -- so two hosts (this container, the GPU box)
produce bit-identical float32 tensors regardless of numpy/CPU SIMD dispatch.

Normals are Irwin-Hall(12): the sum of twelve 16-bit uniforms minus 6 (variance exactly 1,
support [-6, 6], kurtosis 2.9).  This is what the synthetic checkpoints, the golden vectors and
``bench.py``'s synthetic batches are drawn from.
"""

from __future__ import annotations

import numpy as np

__all__ = ["key_for", "uniform", "normal", "randint"]

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def key_for(seed: int, name: str) -> np.uint64:
    k = np.array([(int(seed) * 0x9E3779B97F4A7C15 + _fnv1a64(name)) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
    return _mix(_mix(k))[0]


def _draw(key: np.uint64, n: int, lane: int, lanes: int) -> np.ndarray:
    """n raw 64-bit words: word i of lane `lane` (of `lanes` words per element)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(lanes) + np.uint64(lane)
        return _mix(key + (ctr + np.uint64(1)) * _GOLDEN)


def uniform(seed: int, name: str, shape, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """U[lo, hi) as float64 (53 random bits per element)."""
    n = int(np.prod(shape, dtype=np.int64))
    w = _draw(key_for(seed, name), n, 0, 1)
    u = (w >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).reshape(shape)


def normal(seed: int, name: str, shape) -> np.ndarray:
    """Approximately N(0, 1) as float64: Irwin-Hall(12) from three 64-bit words per element."""
    n = int(np.prod(shape, dtype=np.int64))
    key = key_for(seed, name)
    acc = np.zeros(n, dtype=np.int64)
    mask = np.uint64(0xFFFF)
    for lane in range(3):
        w = _draw(key, n, lane, 3)
        for sh in (0, 16, 32, 48):
            acc += ((w >> np.uint64(sh)) & mask).astype(np.int64)
    # sum of 12 values (k + 0.5) / 65536, minus 6
    return ((acc.astype(np.float64) + 6.0) * (1.0 / 65536.0) - 6.0).reshape(shape)


def randint(seed: int, name: str, shape, high: int) -> np.ndarray:
    """Integers in [0, high) as int64 (modulo bias irrelevant for test data)."""
    n = int(np.prod(shape, dtype=np.int64))
    w = _draw(key_for(seed, name), n, 0, 1)
    return (w % np.uint64(high)).astype(np.int64).reshape(shape)
