"""csrc/tap_gemm.h gelu1 (round 4): the GELU epilogue's erf is ONE polynomial branch -- erf(t) = 1 - 2^-(t P7(t)) -- instead of ocml's
erff.  The header's constants are parsed and the function is emulated operation by operation in numpy (every multiply / fma rounded to
fp32 once, 2^x exact then rounded: the hardware's v_exp_f32 is within one ulp of that) against float64: the bounds the header
states, the behaviour beyond the fitted range, and that the result is no worse than torch's own fp32 GELU.  CPU-only."""
import os
import re

import numpy as np
import torch
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def _coefficients():
    src = open(os.path.join(ROOT, "audiocodecs_amd", "csrc", "tap_gemm.h")).read()
    body = src[src.index("__device__ __forceinline__ float gelu1(float v) {"):]
    body = body[: body.index("\n}\n")]
    lead = float(re.search(r"float p = ([-0-9.e+]+)f;", body).group(1))
    rest = [float(m) for m in re.findall(r"p = fmaf\(p, t, ([-0-9.e+]+)f\);", body)]
    assert len(rest) == 7 and "0.70710678118654752440f" in body and "__builtin_amdgcn_exp2f(-(p * t))" in body and "fmaf(hv, er, hv)" in body
    return [lead] + rest          # highest degree first


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _gelu(x):
    cs = _coefficients()
    u = (x * f32(0.70710678118654752440)).astype(np.float32)
    t = np.abs(u)
    with np.errstate(over="ignore", invalid="ignore"):
        p = np.full_like(t, f32(cs[0]))
        for c in cs[1:]:
            p = _fma(p, t, np.full_like(t, f32(c)))
        s = (p * t).astype(np.float32)
        e = np.exp2(-s.astype(np.float64)).astype(np.float32)
    er = np.copysign((f32(1.0) - e).astype(np.float32), u)
    hv = (f32(0.5) * x).astype(np.float32)
    return _fma(hv, er, hv), er, u, s


def test_gelu1_error_bounds_against_float64():
    rng = np.random.default_rng(0)
    x = np.concatenate([np.linspace(-8, 8, 1600001), rng.normal(size=800000) * 2, np.logspace(-8, 1, 100000), -np.logspace(-8, 1, 100000)]).astype(np.float32)
    y, er, u, _ = _gelu(x)
    x64 = x.astype(np.float64)
    ref = 0.5 * x64 * (1.0 + erf(x64 / np.sqrt(2.0)))
    assert np.abs(er.astype(np.float64) - erf(u.astype(np.float64))).max() <= 1.2e-7
    err = np.abs(y - ref)
    assert err.max() <= 6e-7, err.max()
    m = np.abs(ref) > 1e-3
    assert (err[m] / np.abs(ref[m])).max() <= 1e-4
    yt = torch.nn.functional.gelu(torch.from_numpy(x)).numpy()             # the reference's own fp32 evaluation
    assert err.max() <= np.abs(yt - ref).max()
    assert np.abs(y - yt).max() <= 1.5e-6                                   # what the swap can move a layer output by


def test_gelu1_beyond_the_fitted_range():
    xx = np.concatenate([np.linspace(5.5, 200, 400001), np.logspace(2, 38, 4000)]).astype(np.float32)
    for sign in (1.0, -1.0):
        x = (sign * xx).astype(np.float32)
        y, er, _, s = _gelu(x)
        assert not np.isnan(s).any() and float(np.nanmin(s)) >= 24.5      # 2^-24.5: erf within 4e-8 of +-1
        assert np.all(np.abs(er) >= 1.0 - 6e-8)
        if sign > 0:
            assert np.all(np.abs(y[np.isfinite(y)] - x[np.isfinite(y)]) <= 1.2e-7 * x[np.isfinite(y)])
        else:
            assert np.all(np.abs(y) <= 1.3e-7 * np.abs(x))               # x (1 + erf) / 2 with 1 + erf <= 2^-23
    y, *_ = _gelu(np.array([np.nan, np.inf, 0.0, -0.0], dtype=np.float32))
    assert np.isnan(y[0]) and y[1] == np.inf and y[2] == 0.0 and y[3] == 0.0
