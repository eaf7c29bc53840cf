"""Accuracy of csrc/tap_gemm.h sin2_f32 (emulated operation by operation in numpy: every fma / mul rounded to fp32 once) against
float64 sin(t)^2, beside the reference's own fp32 evaluation (sin rounded to fp32, then squared).  CPU only."""
import numpy as np
f32, f64 = np.float32, np.float64
def fma(a, b, c): return f32(f64(a) * f64(b) + f64(c))
def mul(a, b): return f32(f64(a) * f64(b))
C = [f32(-0.1666666716337204), f32(0.008333329111337662), f32(-0.00019839312881231308), f32(2.7181215500604594e-06)]
PH, PL, PL2 = f32(float.fromhex("0x1.921fb6p+0")), f32(float.fromhex("-0x1.777a5cp-25")), f32(float.fromhex("-0x1.ee59dap-50"))
def sin2_f32(t):
    k = np.rint(mul(t, f32(0.636619747))).astype(f32)
    r = fma(-k, PH, t); r = fma(-k, PL, r); r = fma(-k, PL2, r)
    z = mul(r, r)
    q = fma(C[3], z, C[2]); q = fma(q, z, C[1]); q = fma(q, z, C[0])
    s = fma(mul(r, z), q, r)
    s2 = mul(s, s)
    return np.where(k.astype(np.int64) & 1, f32(1) - s2, s2).astype(f32)
rng = np.random.default_rng(0)
for scale in (1.0, 10.0, 300.0, 5000.0, 20000.0):
    t = (rng.standard_normal(2_000_000) * scale).astype(f32)
    t = t[np.abs(t) < 32768]
    ref = np.sin(t.astype(f64)) ** 2
    e = np.abs(sin2_f32(t).astype(f64) - ref)
    s32 = np.sin(t.astype(f64)).astype(f32)
    e32 = np.abs(mul(s32, s32).astype(f64) - ref)
    print(f"|t| ~ {scale:7.0f}: sin2_f32 max abs err {e.max():.3e}, rms {np.sqrt((e**2).mean()):.3e} | fp32 sin, squared: max {e32.max():.3e}, rms {np.sqrt((e32**2).mean()):.3e}")
