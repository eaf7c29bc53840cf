// rb_stream6: the fused 64-channel SEANet residual block (rb_fused6.h: same block, same split16 arithmetic, same accumulation order)
//     y = shortcut(x) + conv_1x1(ELU(conv_k3(ELU(x))))                                   [HF] modeling_encodec.py:252-282
// rebuilt for FOUR waves per SIMD (round 6).  rb_fused6 keeps a wave's share of the weights in registers (144 of 250), runs at two waves
// per SIMD and spends its time in exposed LDS round trips and vector issue (profiles/r5_thin_stage_notes.md).  Here
//   * the fragment-packed weights (48 KB) live in LDS, loaded once per workgroup (1024 threads = 16 waves, one workgroup per CU);
//   * ONE WAVE = ONE STREAM: a wave walks a segment of one clip in tiles of 16 time rows and owns a 5 KB slab; there is no workgroup
//     barrier after the weights have landed and no cross-wave traffic;
//   * a lane loads the rows in MFMA B-operand shape -- lane (li = lane & 15, kq = lane >> 4) holds, of time row li and every 32-channel
//     k-step kc, the channels  32 kc + 4 kq + {0..3}  and  32 kc + 16 + 4 kq + {0..3}  (two 16-byte loads; four lanes cover 64 contiguous
//     bytes).  The k order inside a dot product is free as long as both operands use the same one, so the weight images are packed
//     offline in THAT order (core.h rb6: the *p images).  It is also the order the accumulators of the transposed tile come in (a lane's
//     4 values of output-channel tile c are channels 16 c + 4 kq + {0..3}), hence
//       - the raw rows are split in registers and ARE the shortcut conv's operand: no raw planes in LDS;
//       - the hidden activation goes accumulator -> ELU -> split -> operand registers: it never touches LDS;
//       - the identity shortcut (Mimi) adds the registers the rows were loaded into: no second read;
//     only ELU(x) passes through LDS, because the k3 conv reads it at three row offsets.  The two halo rows of a tile are the previous
//     tile's last rows, copied inside the slab; a segment's first tile loads them (reflect / zero rule of rb_fused6.h);
//   * slab layout [kq 4][80 units of 16 B]: unit = plane 36 + kc 18 + row (18 rows).  A ds_read_b128 is served in groups of 16 lanes
//     taken from two kq values (MI355X_MICROARCH.md, LDS); with the kq blocks a multiple of 16 units apart and rows one unit apart every
//     group covers the 64 banks exactly once at any tap offset.  Weight fragments are lane-linear (1 KB per wave read).
// Accumulation order over k-steps and partial products is rb_fused6's; inside one MFMA the 32 products arrive in the permuted channel order.
#pragma once
#include "rb_params6.h"

namespace ac {

template <bool SC>
struct Rs6Cfg {
    static constexpr int C = 64, HC = 32, WAVES = 16, ROWS = 16;
    static constexpr int KSA = 6, KSB = 1 + (SC ? 2 : 0);
    static constexpr int WA_BYTES = 2 * KSA * 2 * 1024;          // [n-tile 2][k-step 6][plane 2] x 1 KB
    static constexpr int WB_BYTES = 4 * KSB * 2 * 1024;          // [n-tile 4][k-step KSB][plane 2] x 1 KB
    static constexpr int CONST_FLOATS = 32 + 32 + 64 + 64;       // b3, winv3, bf, winvf
    static constexpr int KQ_UNITS = 80, PL_UNITS = 36, KC_UNITS = 18;
    static constexpr int SLAB_BYTES = 4 * KQ_UNITS * 16;
    static constexpr size_t lds_bytes = (size_t)WA_BYTES + WB_BYTES + CONST_FLOATS * 4 + (size_t)WAVES * SLAB_BYTES;
};

struct Hl8 { f16x8 hi, lo; };
// elu4 (tap_gemm.h: v > 0 ? v : __expf(v) - 1) with the multiply by log2(e) and the -1 as packed instructions: the same three
// roundings per element (v_mul, v_exp, v_add), two vector instructions fewer per pair -- the library is built without the SLP
// vectoriser (build.sh), so packed arithmetic has to be spelled
__device__ __forceinline__ f32x4 elu4p(const f32x4 v) {
    const f32x4 a = v * 1.44269504088896340736f;
    const f32x4 t = f32x4{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y), __builtin_amdgcn_exp2f(a.z), __builtin_amdgcn_exp2f(a.w)} - 1.0f;
    return f32x4{v.x > 0.f ? v.x : t.x, v.y > 0.f ? v.y : t.y, v.z > 0.f ? v.z : t.z, v.w > 0.f ? v.w : t.w};
}
// largest FINITE magnitude of a lane's 16 output values as amax_acc computes it (split16.h: NaN and inf do not enter), for the price of
// a float maximum: fmax ignores NaN, and the one case it gets wrong -- an inf among the values -- is caught and redone the long way
__device__ __forceinline__ unsigned amax16(const f32x4 (&v)[4]) {
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[c].x), __builtin_fabsf(v[c].y)));
        m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[c].z), __builtin_fabsf(v[c].w)));
    }
    unsigned mb = __float_as_uint(m);
    if (mb >= 0x7f800000u) {
        mb = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) amax_acc4(mb, v[c]);
    }
    return mb;
}
// 8 fp32 (two float4) x scale -> hi / lo fp16 operand registers; the arithmetic of split16_store4s (split16.h), element for element
__device__ __forceinline__ Hl8 split16_regs8(const f32x4 a, const f32x4 b, const float s) {
    const f16x4_t ha = __builtin_convertvector(a * s, f16x4_t), hb = __builtin_convertvector(b * s, f16x4_t);
    Hl8 r;
    r.hi = f16x8{ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
    r.lo = f16x8{(_Float16)__builtin_fmaf(a.x, s, -(float)ha.x), (_Float16)__builtin_fmaf(a.y, s, -(float)ha.y),
                 (_Float16)__builtin_fmaf(a.z, s, -(float)ha.z), (_Float16)__builtin_fmaf(a.w, s, -(float)ha.w),
                 (_Float16)__builtin_fmaf(b.x, s, -(float)hb.x), (_Float16)__builtin_fmaf(b.y, s, -(float)hb.y),
                 (_Float16)__builtin_fmaf(b.z, s, -(float)hb.z), (_Float16)__builtin_fmaf(b.w, s, -(float)hb.w)};
    return r;
}
// acc += W x^T over one k-step: lo hi, hi lo, hi hi (mma6's order)
__device__ __forceinline__ f32x4 mma16(const f16x8 wh, const f16x8 wl, const f16x8 xh, const f16x8 xl, f32x4 v) {
    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, v, 0, 0, 0);
    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, v, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, v, 0, 0, 0);
}

// p.w3f / p.wff: the PERMUTED images (ResBlockPlan::w3p_off / wfp_off); p.nseg segments of p.seg_rows rows (a multiple of 16) per clip
// YR / YE: the raw / the ELU'd flavour of the output is stored (p.y / p.y_elu)
template <bool SC, bool YR, bool YE>
__global__ __launch_bounds__(1024) void rb_stream6_kernel(const RbFused6Params p) {
    using Cfg = Rs6Cfg<SC>;
    constexpr int KSB = Cfg::KSB;
#if defined(AC_DEVELOPER) && defined(RS6_ABL)      // timing ablations of developer builds (wrong results): 1 no stage A, 2 no stage B, 4 no staging
    constexpr int ABL = RS6_ABL;                   // arithmetic, 8 no stores, 16 no loads, 32 no output ELU
#else
    constexpr int ABL = 0;
#endif
#ifndef RS6_LDAUX
#define RS6_LDAUX 0
#endif
#ifndef RS6_STAUX
#define RS6_STAUX 0
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rs[];
    unsigned char* Wa = smem_rs;
    unsigned char* Wb = Wa + Cfg::WA_BYTES;
    float* Cs = reinterpret_cast<float*>(Wb + Cfg::WB_BYTES);      // b3[32] winv3[32] bf[64] winvf[64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    unsigned char* slab = reinterpret_cast<unsigned char*>(Cs + Cfg::CONST_FLOATS) + wave * Cfg::SLAB_BYTES;

    // ---- weights and per-channel constants -> LDS (once per workgroup)
    for (int i = tid; i < Cfg::WA_BYTES / 16; i += 1024) reinterpret_cast<u32x4_t*>(Wa)[i] = reinterpret_cast<const u32x4_t*>(p.w3f)[i];
    for (int i = tid; i < Cfg::WB_BYTES / 16; i += 1024) reinterpret_cast<u32x4_t*>(Wb)[i] = reinterpret_cast<const u32x4_t*>(p.wff)[i];
    if (tid < 32) { Cs[tid] = p.b3[tid]; Cs[32 + tid] = p.winv3[tid]; }
    else if (tid < 96) { Cs[64 + tid - 32] = p.bf[tid - 32]; Cs[128 + tid - 32] = p.winvf[tid - 32]; }
    __syncthreads();

    const unsigned char* wa_l = Wa + lane * 16;
    const unsigned char* wb_l = Wb + lane * 16;
    unsigned char* sl = slab + (kq * Cfg::KQ_UNITS + li) * 16;    // this lane's unit of slab row 0, plane 0, kc 0
    const float* c_l = Cs + 4 * kq;
    auto unit = [](int pl, int kc, int row) { return (pl * Cfg::PL_UNITS + kc * Cfg::KC_UNITS + row) * 16; };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int clip_bytes = p.L * 256;
    const int total = p.B * p.nseg;

    for (int sg = blockIdx.x * Cfg::WAVES + wave; sg < total; sg += gridDim.x * Cfg::WAVES) {
        const int b = sg / p.nseg;
        const int t_beg = (sg - b * p.nseg) * p.seg_rows;
        const int t_end = t_beg + p.seg_rows < p.L ? t_beg + p.seg_rows : p.L;
        const long long ob = (long long)b * p.L * 64;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xr + ob), 0, clip_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y ? p.y + ob : nullptr), 0, p.y ? clip_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_elu ? p.y_elu + ob : nullptr), 0, p.y_elu ? clip_bytes : 0, 0x00020000);
        const Rb16Scale cs = rb16_scale(*amax_at(p.amax_in, b), p.hb0, p.hb1);
        unsigned omax = 0;

        // rows t .. t + 15 in operand shape: r[kc][h] = channels 32 kc + 16 h + 4 kq + {0..3} of row t + li (rows past the clip: zeros)
        auto request = [&](int t, f32x4 (&r)[2][2]) {
            const int row = t + li;
            const int ro = row < p.L && !(ABL & 16) ? row * 256 + kq * 16 : 0x7fff0000;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (ABL & 768) {   // address-pattern probes for the loads: 256: 8 rows x 128 B per instruction, 512: 4 rows x 256 B
                        const int c = kc * 2 + h;
                        const int r2 = (ABL & 256) ? t + (li & ~1) + (c & 1) : t + (li & ~3) + c;
                        const int o2 = (ABL & 256) ? ((c >> 1) * 2 + (li & 1)) * 64 + kq * 16 : (li & 3) * 64 + kq * 16;
                        r[kc][h] = bufload16(rs, r2 < p.L ? r2 * 256 + o2 : 0x7fff0000, 0);
                    } else
                    r[kc][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ro + kc * 128 + h * 64, 0, RS6_LDAUX));
                }
        };
        // ELU + split of one row set -> slab rows `row0 + li` (both planes)
        auto stage_xe = [&](const f32x4 (&r)[2][2], int row0_bytes) {
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const Hl8 e = (ABL & 4) ? Hl8{__builtin_bit_cast(f16x8, r[kc][0]), __builtin_bit_cast(f16x8, r[kc][1])} : split16_regs8(elu4p(r[kc][0]), elu4p(r[kc][1]), cs.sx);
                *reinterpret_cast<f16x8*>(sl + row0_bytes + unit(0, kc, 0)) = e.hi;
                *reinterpret_cast<f16x8*>(sl + row0_bytes + unit(1, kc, 0)) = e.lo;
            }
        };

        // the two halo rows in front of row t (reflect / zeros: rb_fused6.h load_tile), lanes li < 2
        auto request_halo = [&](int t, f32x4 (&r)[2][2]) {
            int j = t - 2 + li;
            if (p.pad == PAD_REFLECT) j = j < 0 ? -j : (j >= p.Lp ? 2 * (p.Lp - 1) - j : j);
            const int ho = li < 2 && j >= 0 && j < p.L ? j * 256 + kq * 16 : 0x7fff0000;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int h = 0; h < 2; ++h) r[kc][h] = bufload16(rs, ho + kc * 128 + h * 64, 0);
        };

        f32x4 rx[2][2];                     // the tile's raw rows (identity shortcut: kept until the output)
        Hl8 xr[2];                          // ... split with the second stage's scale: the shortcut conv's operand
        {   // ---- first tile of the segment: its rows and the two halo rows
            request(t_beg, rx);
            f32x4 rh[2][2];
            request_halo(t_beg, rh);
            if (li < 2) stage_xe(rh, 0);
            stage_xe(rx, 2 * 16);
            if (SC) { xr[0] = split16_regs8(rx[0][0], rx[0][1], cs.sb); xr[1] = split16_regs8(rx[1][0], rx[1][1], cs.sb); }
        }

        for (int t = t_beg; t < t_end; t += 16) {
            const bool has_next = t + 16 < t_end;
            f32x4 rn[2][2];
            request(has_next ? t + 16 : 0x3fffff00, rn);           // (no next tile: every row out of range, no memory access)
            __builtin_amdgcn_sched_barrier(0);                     // the requests stay HERE: a tile ahead of their use

            // ---- stage A: hidden = ELU(conv_k3(xe) + b3); M = 32 hidden channels (2 tiles), N = this tile's 16 rows, K = 3 taps x 64
            f32x4 accA[2] = {zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < ((ABL & 1) ? 0 : 6); ++ks) {
                const int j = ks >> 1, kc = ks & 1;
                const f16x8 xh = *reinterpret_cast<const f16x8*>(sl + unit(0, kc, j));
                const f16x8 xl = *reinterpret_cast<const f16x8*>(sl + unit(1, kc, j));
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(wa_l + ((c * 6 + ks) * 2 + 0) * 1024);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(wa_l + ((c * 6 + ks) * 2 + 1) * 1024);
                    accA[c] = mma16(wh, wl, xh, xl, accA[c]);
                }
            }
            // hidden activation: accumulator -> true units -> ELU -> split -> the second stage's operand (element e < 4: tile 0, else tile 1)
            Hl8 hf;
            {
                f32x4 hv[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x4 b3v = *reinterpret_cast<const f32x4*>(c_l + 16 * c);
                    const f32x4 iv = *reinterpret_cast<const f32x4*>(c_l + 32 + 16 * c) * cs.ix;
                    const f32x4 v = accA[c];
                    hv[c] = elu4p(f32x4{__fmaf_rn(v.x, iv.x, b3v.x), __fmaf_rn(v.y, iv.y, b3v.y), __fmaf_rn(v.z, iv.z, b3v.z), __fmaf_rn(v.w, iv.w, b3v.w)});
                }
                hf = split16_regs8(hv[0], hv[1], cs.sb);
            }
            // ---- stage B: y = [W1 | Ws] [hidden | x]^T; M = 64 output channels (4 tiles), K = 32 hidden (+ 64 raw channels)
            f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < ((ABL & 2) ? 0 : KSB); ++ks) {
                const f16x8 xh = ks == 0 ? hf.hi : xr[ks - 1 > 0 ? 1 : 0].hi;
                const f16x8 xl = ks == 0 ? hf.lo : xr[ks - 1 > 0 ? 1 : 0].lo;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(wb_l + ((c * KSB + ks) * 2 + 0) * 1024);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(wb_l + ((c * KSB + ks) * 2 + 1) * 1024);
                    acc[c] = mma16(wh, wl, xh, xl, acc[c]);
                }
            }
            // ---- output values: lane (li, kq) holds channels 16 c + 4 kq .. + 3 of time row t + li
            const int row = t + li;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 bfv = *reinterpret_cast<const f32x4*>(c_l + 64 + 16 * c);
                const f32x4 iv = *reinterpret_cast<const f32x4*>(c_l + 128 + 16 * c) * cs.ib;
                f32x4 v = acc[c];
                v = f32x4{__fmaf_rn(v.x, iv.x, bfv.x), __fmaf_rn(v.y, iv.y, bfv.y), __fmaf_rn(v.z, iv.z, bfv.z), __fmaf_rn(v.w, iv.w, bfv.w)};
                if (!SC) {                                         // identity shortcut: x + block(x); acc tile c = the rows' load (c >> 1, c & 1)
                    const f32x4 xv = rx[c >> 1][c & 1];
                    v = f32x4{__fadd_rn(xv.x, v.x), __fadd_rn(xv.y, v.y), __fadd_rn(xv.z, v.z), __fadd_rn(xv.w, v.w)};
                }
                acc[c] = v;
            }
            // ---- the next tile is staged BEFORE this tile's stores are issued (the wait for its rows must not cover the stores):
            //      halo = this tile's last two rows, copied inside the slab (lanes 0..31: one 16-byte unit each)
            // (unconditionally: behind the segment's last tile the rows are zeros and nobody reads the slab again.  Under `if (has_next)`
            //  hipcc sinks the REQUEST into the conditional block, next to its use, and every tile waits out a full HBM round trip)
            {
                if (lane < 32) {
                    unsigned char* hp = slab + ((lane >> 3) * Cfg::KQ_UNITS + ((lane >> 2) & 1) * Cfg::PL_UNITS + ((lane >> 1) & 1) * Cfg::KC_UNITS + (lane & 1)) * 16;
                    const u32x4_t hv = *reinterpret_cast<const u32x4_t*>(hp + 16 * 16);
                    *reinterpret_cast<u32x4_t*>(hp) = hv;
                }
                stage_xe(rn, 2 * 16);
                if (SC) { xr[0] = split16_regs8(rn[0][0], rn[0][1], cs.sb); xr[1] = split16_regs8(rn[1][0], rn[1][1], cs.sb); }
            }
            // ---- stores
            const int orow = row < p.L && !(ABL & 8) ? row * 256 + kq * 16 : 0x7fff0000;       // rows outside the clip: out of range, dropped
            {
                const unsigned tm = amax16(acc);
                omax = row < p.L && tm > omax ? tm : omax;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (ABL & 192) {       // address-pattern probes (garbage placement): 64: 8 rows x 128 B per instruction, 128: 4 rows x 256 B
                    const int r2 = (ABL & 64) ? t + (li & ~1) + (c & 1) : t + (li & ~3) + c;
                    const int o2 = (ABL & 64) ? ((c >> 1) * 2 + (li & 1)) * 64 + kq * 16 : (li & 3) * 64 + kq * 16;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[c]), re, r2 < p.L ? r2 * 256 + o2 : 0x7fff0000, 0, 0);
                    continue;
                }
                if (YR) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[c]), ry, orow + c * 64, 0, RS6_STAUX);
                if (YE) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (ABL & 32) ? acc[c] : elu4p(acc[c])), re, orow + c * 64, 0, RS6_STAUX);
            }
            if (!SC) {
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                    for (int h = 0; h < 2; ++h) rx[kc][h] = rn[kc][h];
            }
        }
        if (p.amax_out) amax_flush(omax, amax_at(p.amax_out, b));
    }
}

}  // namespace ac
