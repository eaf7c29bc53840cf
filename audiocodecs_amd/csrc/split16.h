// split16: fp32-fidelity products on the fp16 matrix pipe with TWO planes per operand and THREE partial products.
//
//   x * 2^s = hi + lo,   hi = fp16_rn(x * 2^s),   lo = fp16_rn(x * 2^s - hi)            (|lo| <= 2^-12 |x 2^s|)
//   a * b  ~=  (a_hi b_hi + a_hi b_lo + a_lo b_hi) * 2^-(sa + sb)                          (a_lo b_lo <= 2^-22 |a b| dropped, worst case)
//
// hi carries 11 significand bits; the residual of that round-to-nearest is below half an ulp of hi (up to 2^-11 |hi| just above a
// power of two, i.e. up to 12 significant bits below hi's last) and lo keeps its leading 11: WORST case the pair represents x to
// 2^-23 relative (typically 2^-24, the precision of the fp32 value) wherever lo is a normal fp16 number, and to 2^-25 ABSOLUTE (of
// the scaled value) below that: v_mfma_f32_*_f16 keeps fp16 denormal inputs (tools/ubench/mfma_f16_probe.hip).  Products of fp16
// values are exact in the fp32 accumulator; the dropped lo lo term is at most 2^-11 * 2^-11 = 2^-22 of |a b| (both residuals at
// their worst), typically 2^-24 or less.  So the arithmetic is fp32-GRADE, not fp32-equal -- measured on K = 1536 dot products of
// random data, rms error / rms value: fp32 FMA chain 1.8e-7, this 1.9e-7 (the removed three-plane bf16 split, 6 products: 1.9e-7).
//
// What fp16 lacks is range, so every operand carries a power-of-two scale (multiplying by it is exact):
//   * weights: one scale per OUTPUT CHANNEL, chosen offline from the row's largest magnitude; 2^-s comes back in the epilogue;
//   * activations: one scale per CLIP and tensor, from the tensor's largest magnitude ("amax", bits of |x| as an unsigned
//     int -- ordered like the values, NaN above everything) which the PRODUCING kernel leaves in an amax slot by atomicMax
//     (one per wave, skipped when the slot already holds more), or a bound derived from it (ELU never grows a magnitude;
//     |conv(x)| <= |b| + ||w||_1 amax(x) inside the fused blocks; LSTM output <= 1 + amax(skip); |re|, |im| <= 100 behind the
//     clamped magnitude of WavTokenizer's head).
//     Scaled magnitudes stay below 2^15, so elements down to 2^-16 of the clip's largest keep fp32-grade RELATIVE
//     precision and everything smaller an absolute error of 2^-40 of the largest -- far below the rounding of the fp32
//     accumulation it feeds.  inf / NaN elements do not enter the amax: they convert to fp16 inf / NaN and propagate, the
//     clip's finite part keeps its scale.
// A tensor whose producer does not report an amax (normalisation kernels, codebook sums) gets one from amax_kernel (one extra
// read) -- never a guess; whoever rewrites a tensor in place drops its slot (Act::amax on the host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ac {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef float s16_f32x4 __attribute__((ext_vector_type(4)));

// scale exponent for a tensor whose largest magnitude has the bits `amax_bits` (sign cleared): |x| 2^s < 2^15
// (clamped to +-lim: 80 for activations, 40 for weight rows -- the product of the two inverse scales stays a normal fp32)
__host__ __device__ __forceinline__ int s16_exponent(unsigned amax_bits, int lim = 80) {
    const int e = (int)((amax_bits >> 23) & 0xffu);
    if (e == 255) return 0;                 // (a bound that overflowed: scale 1)
    int s = 141 - e;                        // |x| < 2^(e - 126)  ->  |x| 2^s < 2^15
    return s > lim ? lim : (s < -lim ? -lim : s);
}
__host__ __device__ __forceinline__ float s16_pow2(int s) {
    union { unsigned u; float f; } c;
    c.u = (unsigned)(s + 127) << 23;
    return c.f;
}

// A slot holds one word per clip, AMAX_STRIDE words (one 128-byte line) apart: the words of different clips are polled and
// raised by different workgroups at the same time, and 64 neighbouring words would queue all of that on two L2 lines
constexpr int AMAX_STRIDE = 32;
__host__ __device__ __forceinline__ unsigned* amax_at(unsigned* slot, int b) { return slot + (long long)b * AMAX_STRIDE; }
__host__ __device__ __forceinline__ const unsigned* amax_at(const unsigned* slot, int b) { return slot + (long long)b * AMAX_STRIDE; }

// |v| bits into a running maximum: the largest FINITE magnitude -- inf / NaN elements stay what they are in fp16 and do not
// enter the scale (v_cmp_class + v_cndmask |v| + v_max_u32: three instructions per value)
__device__ __forceinline__ void amax_acc(unsigned& mx, float v) {
    const unsigned b = __builtin_amdgcn_classf(v, 0x1F8) ? __float_as_uint(__builtin_fabsf(v)) : 0u;     // 0x1F8: +-normal, +-subnormal, +-0
    mx = b > mx ? b : mx;
}
__device__ __forceinline__ void amax_acc4(unsigned& mx, const s16_f32x4 v) {
    amax_acc(mx, v.x); amax_acc(mx, v.y); amax_acc(mx, v.z); amax_acc(mx, v.w);
}
// wave maximum -> slot (one atomic per wave at most; none when the slot already holds at least as much)
__device__ __forceinline__ void amax_flush(unsigned mx, unsigned* slot) {
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)mx, o);
        mx = t > mx ? t : mx;
    }
    if ((threadIdx.x & 63) == 0 && mx > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, mx);
}

// acc = a * b + acc as ONE v_fma_f32 that no vectoriser can pack.  The fused encoder front's stem returned wrong, run-to-run different
// values when SLP vectorisation turned its scalar FMAs into v_pk_fma_f32 with op_sel broadcasts (profiles/r3_pk_fma_hazard.md: only
// with a second wave on the SIMD, mechanism not established).  -fno-slp-vectorize (build.sh) keeps the rest of the library free of them;
// the stem does not depend on the flag, and csrc/check_isa.sh fails the build if a packed FMA appears in the fused chains.
__device__ __forceinline__ float fma_pinned(float a, float b, float acc) {
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}

// 4 fp32 values x scale -> hi / lo fp16 planes, 8 bytes each at element offset `o` (planes `plane` elements apart):
//   hi = fp16(v s),  lo = fp16(fma(v, s, -hi))   -- v_fma_mixlo/mixhi_f16 does the fma on the fp16 `hi` and the rounding of the
// result in ONE instruction: two instructions per value in all (v_pk_mul, v_cvt_pk, 2 x v_fma_mix per pair)
__device__ __forceinline__ void split16_store4s(const s16_f32x4 v, const float s, void* p0, int plane, int o) {
    const f16x4_t hi = __builtin_convertvector(v * s, f16x4_t);
    f16x4_t lo;
    lo.x = (_Float16)__builtin_fmaf(v.x, s, -(float)hi.x);
    lo.y = (_Float16)__builtin_fmaf(v.y, s, -(float)hi.y);
    lo.z = (_Float16)__builtin_fmaf(v.z, s, -(float)hi.z);
    lo.w = (_Float16)__builtin_fmaf(v.w, s, -(float)hi.w);
    *reinterpret_cast<f16x4_t*>(reinterpret_cast<_Float16*>(p0) + o) = hi;
    *reinterpret_cast<f16x4_t*>(reinterpret_cast<_Float16*>(p0) + plane + o) = lo;
}

}  // namespace ac
