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

// Developer modes.  The product library (build.sh) is compiled WITHOUT AC_DEVELOPER: the run-time timing / fault-injection words
// (RbFused6Params::dbg, LstmPersistParams::dbg) then read as 0 inside the kernels, ac_debug_set refuses their keys, ac_finalize does not
// look at AC_RB6_DBG / AC_LSTM_DBG, and the compile-time trace / ablation switches below are an error.  The developer library
// (libaudiocodecs_amd_dev.so, same build.sh) carries them for the fault-injection tests and tools/experiments.
#ifdef AC_DEVELOPER
#define AC_DEV_MODE(word, bits) ((word) & (bits))
#else
#define AC_DEV_MODE(word, bits) 0
#if defined(T6_TRACE) || defined(TAP4_TRACE) || defined(T6_ABL_NOALOAD) || defined(T6_ABL_NOSTORE) || defined(T6_ABL_NOBLOAD) || defined(T6_ABL_NOMFMA) || \
    defined(T8_ABL_NODMA) || defined(T8_ABL_NOALOAD) || defined(T8_ABL_NOSTORE) || defined(T8_ABL_NOMFMA) || defined(RS6_ABL) || defined(LSTM_DBG_SKIP) || defined(RVQ16_COND_LOADS)
#error "trace / ablation switches (wrong results or extra stores) need -DAC_DEVELOPER: build a side library with tools/devbuild.sh, the product does not carry them"
#endif
#endif

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
// Maximum over aligned groups of LANES lanes (2 .. 64, a power of two), delivered to EVERY lane of the group -- what the xor butterfly of
// __shfl_xor computes, without its LDS round trips: __shfl_xor is ds_bpermute_b32 (~120+ cycles each, waited for one by one: five of them per
// row and store-loop iteration were ~0.6 k of the ~2.5 k cycles an iteration of the row-mode epilogue took, round-5 trace of Mimi's linear
// layers).  Inside a row of 16 lanes the exchanges are DPP modifiers of the v_max itself (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror: a max does not care that the last two are reflections rather than xors); only the 32- and 64-lane steps cross DPP rows and
// keep their bpermute.  Integer maxima only: a float SUM would change its pairing order with the reflections.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
    return t > v ? t : v;
}
template <int LANES>
__device__ __forceinline__ unsigned group_max_u32(unsigned v) {
    static_assert(LANES >= 1 && LANES <= 64 && (LANES & (LANES - 1)) == 0, "power of two up to the wave");
    if (LANES >= 2) v = dpp_max_u32<0xB1>(v);      // quad_perm [1, 0, 3, 2]
    if (LANES >= 4) v = dpp_max_u32<0x4E>(v);      // quad_perm [2, 3, 0, 1]
    if (LANES >= 8) v = dpp_max_u32<0x141>(v);     // row_half_mirror
    if (LANES >= 16) v = dpp_max_u32<0x140>(v);    // row_mirror
    if (LANES >= 32) { const unsigned t = (unsigned)__shfl_xor((int)v, 16); v = t > v ? t : v; }
    if (LANES >= 64) { const unsigned t = (unsigned)__shfl_xor((int)v, 32); v = t > v ? t : v; }
    return v;
}

// wave maximum -> slot (one atomic per wave at most; none when the slot already holds at least as much)
__device__ __forceinline__ void amax_flush(unsigned mx, unsigned* slot) {
    mx = group_max_u32<64>(mx);
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
