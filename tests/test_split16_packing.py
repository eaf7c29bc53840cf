"""CPU check of the host-side split16 packer (csrc/split16.h; no GPU): the two fp16 planes of a weight row and its power-of-two
scale, against numpy's float16 (round-to-nearest-even, denormals kept)."""
import ctypes as C

import numpy as np
import pytest

from test_native_abi import _built


def _split(row):
    L = _built().lib()
    row = np.ascontiguousarray(row, dtype=np.float32)
    hi = np.zeros(row.size, np.uint16)
    lo = np.zeros(row.size, np.uint16)
    s = L.ac_debug_split_row(row.ctypes.data_as(C.c_void_p), row.size, hi.ctypes.data_as(C.c_void_p), lo.ctypes.data_as(C.c_void_p))
    return s, hi.view(np.float16), lo.view(np.float16)


@pytest.mark.parametrize("scale", [1.0, 1e-4, 3e4, 1e-12])
def test_planes_match_numpy_float16(scale):
    rng = np.random.default_rng(5)
    row = (rng.standard_normal(4096) * scale).astype(np.float32)
    row[::97] *= 1e-3          # small entries beside large ones
    row[5] = 0.0
    s, hi, lo = _split(row)
    amax = float(np.abs(row).max())
    assert amax * 2.0**s < 2.0**15 and (amax * 2.0 ** (s + 1) >= 2.0**15 or s == 40)      # the tightest power of two (clamped at 2^40)
    ws = (row.astype(np.float64) * 2.0**s).astype(np.float32)                               # exact: a power of two
    ref_hi = ws.astype(np.float16)
    ref_lo = (ws - ref_hi.astype(np.float32)).astype(np.float16)
    assert np.array_equal(hi.view(np.uint16), ref_hi.view(np.uint16))
    assert np.array_equal(lo.view(np.uint16), ref_lo.view(np.uint16))
    # the pair carries the value to 2^-24 relative where lo is a normal fp16 number, to 2^-25 absolute (scaled) below
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    err = np.abs(rec - ws.astype(np.float64))
    assert np.all(err <= np.maximum(np.abs(ws) * 2.0**-23, 2.0**-25))


def test_rounding_edges():
    # ties to even at the fp16 grid, the largest finite value, denormal lo
    row = np.array([1.0 + 2.0**-11, 1.0 + 3 * 2.0**-11, 65504.0 / 32768.0 * 0.999, 2.0**-20, -(1.0 + 2.0**-12), 1.99993896484375], np.float32)
    s, hi, lo = _split(row)
    ws = (row.astype(np.float64) * 2.0**s).astype(np.float32)
    ref_hi = ws.astype(np.float16)
    assert np.array_equal(hi.view(np.uint16), ref_hi.view(np.uint16))
    assert np.array_equal(lo.view(np.uint16), (ws - ref_hi.astype(np.float32)).astype(np.float16).view(np.uint16))
    assert np.all(np.isfinite(hi.astype(np.float32)))


def test_zero_row_and_argument_check():
    s, hi, lo = _split(np.zeros(32, np.float32))
    assert s == 40 and not hi.view(np.uint16).any() and not lo.view(np.uint16).any()
    assert _built().lib().ac_debug_split_row(None, 4, None, None) == -1
