"""csrc/tap_gemm.h sin2_f32 (the sin^2 inside every Snake activation of the DAC path) -- CPU check of the ALGORITHM and its
constants: the header's constants are parsed, the function is emulated operation by operation in numpy (every fma / mul rounded to
fp32 once, as the device does) and compared with float64 sin(t)^2; the reference's own evaluation (fp32 sin, squared) is the
yardstick.  The device function itself is covered by the DAC parity tests (-m gpu)."""
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(__file__), "..", "audiocodecs_amd", "csrc", "tap_gemm.h")
f32, f64 = np.float32, np.float64


def fma(a, b, c):
    return f32(f64(a) * f64(b) + f64(c))


def mul(a, b):
    return f32(f64(a) * f64(b))


def header_constants():
    src = open(HDR).read()
    body = src[src.index("float sin2_f32(float t)") :]
    body = body[: body.index("\n}\n")]
    hexes = [float.fromhex(h.rstrip("f")) for h in re.findall(r"-?0x[0-9a-fA-F.]+p[+-]?\d+f", body)]
    decs = [float(d.rstrip("f")) for d in re.findall(r"(?<![\w.])-?\d+\.\d+(?:e-?\d+)?f", body)]
    return body, hexes, decs


def test_constants_are_a_three_term_split_of_half_pi():
    body, hexes, decs = header_constants()
    ph, pl, pl2 = hexes[:3]                      # k * (-PH) + t, k * PL + r, k * PL2 + r
    assert ph < 0 < pl and pl2 > 0
    from decimal import Decimal, getcontext

    getcontext().prec = 50
    half_pi = Decimal("3.14159265358979323846264338327950288419716939937510") / 2
    assert abs(half_pi - (Decimal(-ph) - Decimal(pl) - Decimal(pl2))) < Decimal("1e-22")
    assert f32(-ph) == f32(float(half_pi)) and abs(0.636619747 - 2 / np.pi) < 5e-8 and 0.636619747 in decs
    assert "32768.f" in body                     # the large-argument / non-finite path hands over to sinf


def test_emulated_accuracy_against_float64():
    _, hexes, decs = header_constants()
    nph, pl, pl2 = (f32(x) for x in hexes[:3])
    c3, c2, c1, c0 = (f32(x) for x in [d for d in decs if abs(d) < 0.2 and d != 0.636619747][:4])   # Horner order in the source
    assert abs(c0 + 1 / 6) < 1e-6 and abs(c1 - 1 / 120) < 1e-6

    def sin2(t):
        k = np.rint(mul(t, f32(0.636619747))).astype(f32)
        r = fma(k, nph, t)
        r = fma(k, pl, r)
        r = fma(k, pl2, r)
        z = mul(r, r)
        q = fma(c3, z, c2)
        q = fma(q, z, c1)
        q = fma(q, z, c0)
        s = fma(mul(r, z), q, r)
        s2 = mul(s, s)
        return np.where(k.astype(np.int64) & 1, f32(1) - s2, s2).astype(f32)

    rng = np.random.default_rng(0)
    for scale in (1.0, 30.0, 3000.0, 20000.0):
        t = (rng.standard_normal(400_000) * scale).astype(f32)
        t = t[np.abs(t) < 32768]
        ref = np.sin(t.astype(f64)) ** 2
        err = np.abs(sin2(t).astype(f64) - ref)
        s32 = np.sin(t.astype(f64)).astype(f32)
        err32 = np.abs(mul(s32, s32).astype(f64) - ref)
        assert err.max() < 1.5e-7, (scale, err.max())
        assert np.sqrt((err**2).mean()) <= 1.05 * np.sqrt((err32**2).mean()), scale     # rms no worse than fp32 sin, squared
    edge = np.array([0.0, -0.0, np.pi / 4, -np.pi / 4, np.pi / 2, 32767.9], dtype=f32)
    assert np.all(np.abs(sin2(edge).astype(f64) - np.sin(edge.astype(f64)) ** 2) < 1.5e-7)
