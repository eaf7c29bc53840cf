// rb_fused6: the fused SEANet residual block of rb_fused.h (same block, same padding rules, same tile walk)
//     y = shortcut(x) + conv_1x1(ELU(conv_k3(ELU(x))))
// in split-operand arithmetic on the bf16 matrix pipe (tap_gemm6.h: every fp32 value = hi + mid + lo, three exact
// bf16 terms; 6 of the 9 exact partial products, fp32 accumulate).  rb_fused.h runs at 0.45-0.57 of the fp32-MFMA
// peak and is bound by it; the same products cost 0.375x on v_mfma_f32_16x16x32_bf16.
//   * x is read ONCE per tile (raw rows; ELU is applied while staging), split once, and kept in LDS as bf16 planes:
//     Xe = ELU(x) with the 2-row causal halo for the k3 conv, Xr = raw rows for the shortcut conv;
//   * weights: split offline, packed in MFMA fragment order, register-resident for the whole kernel
//       w3f[n-tile of 16][k-step of 32][plane 3][lane 64][8 bf16]      k = tap * C + ci
//       wff[n-tile of 16][k-step of 32][plane 3][lane 64][8 bf16]      k = hidden (zero-padded to 32) | x
//   * the MFMA computes the TRANSPOSED tile (M = output channels from the weight fragment, N = 16 time rows from
//     the x fragment), so a lane's 4 accumulator registers are 4 CONSECUTIVE channels of one time row: the hidden
//     activation goes back to LDS as one 8-byte store per plane, the block output straight to HBM as 16-byte stores
//     (no output tile in LDS, no barrier around it).
// Workgroup = 4 waves, 64 time rows per tile; C = 64: waves split the output channels two ways (their share of the
// weights is 144 VGPRs), C = 32: every wave owns 16 rows and all channels (no barrier between the two stages).
#pragma once
#include <hip/hip_runtime.h>
#include "tap_gemm6.h"
#include "rb_params6.h"

namespace ac {

template <int C, bool SC>
struct Rb6Cfg {
    static constexpr int BM = 64;
    static constexpr int HC = C / 2, HCP = HC < 32 ? 32 : HC;          // hidden width, padded to one k-step
    static constexpr int NSPLIT = C / 32;                               // waves along the output channels
    static constexpr int MW = 4 / NSPLIT;                               // wave groups along time
    static constexpr int MS = BM / MW / 16;                             // 16-row time tiles per wave
    static constexpr int NA = HC / NSPLIT / 16, NB = C / NSPLIT / 16;   // 16-channel tiles per wave (stage A / B)
    static constexpr int KSA = 3 * C / 32;                              // k-steps of stage A
    static constexpr int KSH = HCP / 32, KSB = KSH + (SC ? C / 32 : 0); // k-steps of stage B: hidden, then x
    static constexpr int XP = C + 8, HP = HCP + 8;                      // LDS row pitches in bf16 (16-byte multiples, banks spread)
    static constexpr int XE_ROWS = BM + 2;
    static constexpr int XE_PLANE = XE_ROWS * XP, XR_PLANE = SC ? BM * XP : 0, H_PLANE = BM * HP;
    static constexpr int SLOTS = (XE_ROWS * (C / 4) + 255) / 256;
    static constexpr size_t lds_bytes16 = (size_t)2 * (XE_PLANE + XR_PLANE + H_PLANE) * 2;   // split16.h: two planes
    static constexpr int YP = C + 4;                                    // HEAD: fp32 rows of the ELU'd output tile
    static constexpr size_t lds_bytes16_head = lds_bytes16 + (size_t)BM * YP * 4;
};


// 4 fp32 -> 4 fp16 per plane (split16.h: scaled value = hi + lo), stored as 8 bytes at element offset `o` of each plane.
// (NP: the number of operand planes.  Rounds 1-3 also instantiated 3 bf16 planes / 6 products and a rounded single plane; only
//  split16 remains, the template argument stays in the kernel names the profiles carry.)
template <int NP = 2>
__device__ __forceinline__ void split_store4(const f32x4 v, __bf16* p0, int plane, int o, float scale = 1.f) {
    static_assert(NP == 2, "split16 arithmetic only");
    split16_store4s(v, scale, p0, plane, o);
}

// acc (+)= W x^T over one k-step: lo hi, hi lo, hi hi on the fp16 pipe (small terms first).  w / x: [plane]
template <int NP = 2>
__device__ __forceinline__ f32x4 mma6(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x4 v) {
    static_assert(NP == 2, "split16 arithmetic only");
    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[1]), __builtin_bit_cast(f16x8, x[0]), v, 0, 0, 0);
    v = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, x[1]), v, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, x[0]), v, 0, 0, 0);
}

// split16: two LDS planes and two thirds of the weight registers leave room for one more workgroup per CU where the kernel
// then still fits the register file without spilling -- the identity-shortcut block (Mimi) only; the 1x1-shortcut blocks spill
// 12 / 43 registers at the higher occupancy and run 1.2-1.8x slower
template <int C, bool SC, int NP>
constexpr int rb6_occupancy() { return (C == 64 ? 2 : 3) + ((NP == 2 && C == 64 && !SC) ? 1 : 0); }
template <int C, bool SC, int NP, bool HEADT>
__device__ __forceinline__ void rb_fused6_body(const RbFused6Params& p) {
    using Cfg = Rb6Cfg<C, SC>;
    constexpr int BM = Cfg::BM, HC = Cfg::HC, XP = Cfg::XP, HP = Cfg::HP;
    constexpr int MS = Cfg::MS, NA = Cfg::NA, NB = Cfg::NB, KSA = Cfg::KSA, KSH = Cfg::KSH, KSB = Cfg::KSB;
    constexpr int NSPLIT = Cfg::NSPLIT, SLOTS = Cfg::SLOTS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* Xe = reinterpret_cast<__bf16*>(smem);                  // [3][XE_ROWS][XP]
    constexpr int NPL = NP == 2 ? 2 : 3;                           // planes held in LDS
    __bf16* Xr = Xe + NPL * Cfg::XE_PLANE;                         // [NPL][BM][XP]     (SC only)
    __bf16* Hs = Xr + NPL * Cfg::XR_PLANE;                         // [NPL][BM][HP]
    float* Ys = reinterpret_cast<float*>(Hs + NPL * Cfg::H_PLANE); // [BM][YP]  (HEAD launches only: the launcher sizes the LDS)
    constexpr bool HEAD = HEADT;                                   // (its own kernel: rb_fused6_head_kernel)
    const int tstep = HEAD ? BM - (p.head_k - 1) : BM;             // rows a tile advances by
    const int tback = HEAD ? p.head_k - 1 : 0;                     // rows a tile starts early

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mg = wave / NSPLIT, ng = wave % NSPLIT;
    const int li = lane & 15, kq = lane >> 4;
    const int total = p.B * p.ntiles;
    const int r0 = mg * (BM / Cfg::MW);                            // this wave's first time row in the tile
    const int na0 = ng * NA * 16, nb0 = ng * NB * 16;              // this wave's first output channel (stage A / B)

    // ---- this wave's weight fragments -> registers (once)
    constexpr int WPL = NP == 2 ? 2 : 3;                           // planes in the weight images
    bf16x8 w3r[KSA][NA][3], wfr[KSB][NB][3];
    f32x4 b3v[NA], bfv[NB];
    f32x4 i3v[NA], ifv[NB];                                        // split16: 2^-s of the weight rows
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NA; ++c) {
        b3v[c] = *reinterpret_cast<const f32x4*>(p.b3 + na0 + c * 16 + 4 * kq);
        if (NP == 2) i3v[c] = *reinterpret_cast<const f32x4*>(p.winv3 + na0 + c * 16 + 4 * kq);
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                w3r[ks][c][pl] = *reinterpret_cast<const bf16x8*>(p.w3f + ((((long long)(na0 / 16 + c) * KSA + ks) * WPL + pl) * 64 + lane) * 8);
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        bfv[c] = *reinterpret_cast<const f32x4*>(p.bf + nb0 + c * 16 + 4 * kq);
        if (NP == 2) ifv[c] = *reinterpret_cast<const f32x4*>(p.winvf + nb0 + c * 16 + 4 * kq);
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                wfr[ks][c][pl] = *reinterpret_cast<const bf16x8*>(p.wff + ((((long long)(nb0 / 16 + c) * KSB + ks) * WPL + pl) * 64 + lane) * 8);
    }
    // hidden columns HC .. HCP-1 (C = 32) are K padding: zero once, the matching weight fragments are zero too
    if (HC < Cfg::HCP)
        for (int e = tid; e < NPL * BM * (Cfg::HCP - HC); e += 256) {
            const int pl = e / (BM * (Cfg::HCP - HC)), r = e % (BM * (Cfg::HCP - HC));
            Hs[pl * Cfg::H_PLANE + (r / (Cfg::HCP - HC)) * HP + HC + r % (Cfg::HCP - HC)] = (__bf16)0.f;
        }

    // per-slot constants: slab row / column group of this thread; rows beyond the slab never load (offset out of range)
    int s_row[SLOTS], s_q4[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int e = tid + i * 256;
        s_row[i] = e / (C / 4);
        s_q4[i] = 16 * (e % (C / 4));
    }
    const int clip_bytes = p.L * C * 4;                         // buffer range: loads past the clip return 0, stores are dropped
    f32x4 rx[SLOTS];
    auto load_tile = [&](int tile) {
        const int b = tile / p.ntiles, t0 = (tile % p.ntiles) * tstep - tback;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xr + (long long)b * p.L * C), 0, clip_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            int j = t0 - p.lpad + s_row[i];                      // left pad 2 (causal) or 1 (non-causal k3): reflect ([HF]:157-176) or zeros
            if (p.pad == PAD_REFLECT) j = j < 0 ? -j : (j >= p.Lp ? 2 * (p.Lp - 1) - j : j);
            const bool ok = s_row[i] < Cfg::XE_ROWS && j >= 0 && j < p.L;
            rx[i] = bufload16(rs, ok ? j * (C * 4) + s_q4[i] : 0x7fff0000, 0);
        }
    };
    Rb16Scale sc{1.f, 1.f, 1.f, 1.f};                           // scales of the tile that is staged in LDS
    auto store_tile = [&](int tile) {
        if (NP == 2) sc = rb16_scale(*amax_at(p.amax_in, tile / p.ntiles), p.hb0, p.hb1);
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int row = s_row[i], q = s_q4[i] / 16;
            if (row < Cfg::XE_ROWS) {
                split_store4<NP>(elu4(rx[i]), Xe, Cfg::XE_PLANE, row * XP + 4 * q, sc.sx);
                // rows lpad .. lpad + BM - 1 of the slab are the tile's own rows (source index t0 + row - lpad >= 0)
                if (SC && row >= p.lpad && row < p.lpad + BM) split_store4<NP>(rx[i], Xr, Cfg::XR_PLANE, (row - p.lpad) * XP + 4 * q, sc.sb);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= total) return;
    load_tile(tile);
    store_tile(tile);
    __syncthreads();
    unsigned omax = 0;
    int omax_b = tile / p.ntiles;
    for (; tile < total; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const Rb16Scale cs = sc;                                // this tile's scales (store_tile(next) replaces sc)
        if (next < total && !AC_DEV_MODE(p.dbg, 16)) load_tile(next);     // in flight during both MFMA stages
        // ---- stage A: hidden = ELU(conv_k3(xe) + b3) -> Hs planes
        {
            f32x4 acc[MS][NA];
#pragma unroll
            for (int a = 0; a < MS; ++a)
#pragma unroll
                for (int c = 0; c < NA; ++c) acc[a][c] = NP == 2 ? zero4 : b3v[c];
            if (!AC_DEV_MODE(p.dbg, 1))
#pragma unroll
            for (int ks = 0; ks < KSA; ++ks) {
                const int j = ks / (C / 32), kc = ks % (C / 32);
                bf16x8 xf[MS][3];
#pragma unroll
                for (int a = 0; a < MS; ++a)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        xf[a][pl] = *reinterpret_cast<const bf16x8*>(Xe + pl * Cfg::XE_PLANE + (r0 + a * 16 + li + j) * XP + kc * 32 + 8 * kq);
#pragma unroll
                for (int a = 0; a < MS; ++a)
#pragma unroll
                    for (int c = 0; c < NA; ++c) acc[a][c] = mma6<NP>(w3r[ks][c], xf[a], acc[a][c]);
            }
#pragma unroll
            for (int a = 0; a < MS; ++a)
#pragma unroll
                for (int c = 0; c < NA; ++c) {
                    if (NP == 2) {                               // back to true units: exact power-of-two factors, then the bias
                        const f32x4 iv = i3v[c] * cs.ix;
                        f32x4& v = acc[a][c];
                        v = f32x4{__fmaf_rn(v.x, iv.x, b3v[c].x), __fmaf_rn(v.y, iv.y, b3v[c].y), __fmaf_rn(v.z, iv.z, b3v[c].z), __fmaf_rn(v.w, iv.w, b3v[c].w)};
                    }
                    split_store4<NP>(elu4(acc[a][c]), Hs, Cfg::H_PLANE, (r0 + a * 16 + li) * HP + na0 + c * 16 + 4 * kq, cs.sb);
                }
        }
        if (NSPLIT > 1) lds_barrier();                          // C = 32: a wave reads back only its own rows
        // ---- stage B: y = [W1 | Ws] * [hidden | x]^T + bf
        f32x4 acc[MS][NB];
#pragma unroll
        for (int a = 0; a < MS; ++a)
#pragma unroll
            for (int c = 0; c < NB; ++c) acc[a][c] = NP == 2 ? zero4 : bfv[c];
        if (!AC_DEV_MODE(p.dbg, 2))
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            bf16x8 xf[MS][3];
#pragma unroll
            for (int a = 0; a < MS; ++a)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    xf[a][pl] = ks < KSH ? *reinterpret_cast<const bf16x8*>(Hs + pl * Cfg::H_PLANE + (r0 + a * 16 + li) * HP + ks * 32 + 8 * kq)
                                         : *reinterpret_cast<const bf16x8*>(Xr + pl * Cfg::XR_PLANE + (r0 + a * 16 + li) * XP + (ks - KSH) * 32 + 8 * kq);
#pragma unroll
            for (int a = 0; a < MS; ++a)
#pragma unroll
                for (int c = 0; c < NB; ++c) acc[a][c] = mma6<NP>(wfr[ks][c], xf[a], acc[a][c]);
        }
        lds_barrier();                                          // every wave is done reading the slabs
        // the next tile is staged BEFORE this tile's output stores are issued: the wait for its loads must not
        // cover the stores (the memory counter retires in order)
        if (next < total && !AC_DEV_MODE(p.dbg, 4)) store_tile(next);
        // ---- output: lane (li, kq) holds channels nb0 + 16c + 4kq .. +3 of time row r0 + 16a + li
        if (!AC_DEV_MODE(p.dbg, 8)) {
            const int b = tile / p.ntiles, t0 = (tile % p.ntiles) * tstep - tback;
            if (p.amax_out && b != omax_b) {                     // clip change: hand the finished clip's maximum over
                amax_flush(omax, amax_at(p.amax_out, omax_b));
                omax = 0;
                omax_b = b;
            }
            const long long ob = (long long)b * p.L * C;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y ? p.y + ob : nullptr), 0, p.y ? clip_bytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_elu ? p.y_elu + ob : nullptr), 0, p.y_elu ? clip_bytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xr + ob), 0, clip_bytes, 0x00020000);
#pragma unroll
            for (int a = 0; a < MS; ++a) {
                const int t = t0 + r0 + a * 16 + li;
                const int orow = t >= 0 && t < p.L ? t * (C * 4) : 0x7fff0000;   // rows outside the clip: out of range, dropped
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const int o = orow + (nb0 + c * 16 + 4 * kq) * 4;
                    f32x4 v = acc[a][c];
                    if (NP == 2) {
                        const f32x4 iv = ifv[c] * cs.ib;
                        v = f32x4{__fmaf_rn(v.x, iv.x, bfv[c].x), __fmaf_rn(v.y, iv.y, bfv[c].y), __fmaf_rn(v.z, iv.z, bfv[c].z), __fmaf_rn(v.w, iv.w, bfv[c].w)};
                    }
                    if (!SC) {                                   // identity shortcut: x + block(x)
                        const f32x4 xv = bufload16(rs, o, 0);
                        v = f32x4{__fadd_rn(xv.x, v.x), __fadd_rn(xv.y, v.y), __fadd_rn(xv.z, v.z), __fadd_rn(xv.w, v.w)};
                    }
                    if (HEAD) {      // the final conv's input tile: ELU(y) inside the clip, the conv's zero padding outside
                        *reinterpret_cast<f32x4*>(Ys + (r0 + a * 16 + li) * Cfg::YP + nb0 + c * 16 + 4 * kq) = t >= 0 && t < p.L ? elu4(v) : zero4;
                        continue;
                    }
                    if (p.amax_out && t < p.L) amax_acc4(omax, v);
                    if (p.y) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ry, o, 0, 0);
                    if (p.y_elu) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, elu4(v)), re, o, 0, 0);
                }
            }
            if (HEAD) {
                // sample t0 + tback + tl = sum over taps j, channels c of w[j][c] Ys[tl + j][c] + bias: four lanes per sample, each over a
                // quarter of the channels (all taps), quarters added pairwise, the bias last -- head4_kernel's order (thin.h)
                lds_barrier();
                const int tl = tid >> 2, q = tid & 3;
                float hacc = 0.f;
                for (int j = 0; j < p.head_k; ++j) {
                    const float* yr = Ys + (tl + j < BM ? tl + j : BM - 1) * Cfg::YP + q * (C / 4);     // (rows past the tile: lanes that store nothing)
                    const float* wr = p.head_w + j * C + q * (C / 4);
#pragma unroll
                    for (int g = 0; g < C / 16; ++g) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(yr + 4 * g);
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * g);
                        hacc = fmaf(wv.x, xv.x, hacc); hacc = fmaf(wv.y, xv.y, hacc);
                        hacc = fmaf(wv.z, xv.z, hacc); hacc = fmaf(wv.w, xv.w, hacc);
                    }
                }
                hacc += __shfl_xor(hacc, 1);
                hacc += __shfl_xor(hacc, 2);
                const int ts = t0 + tback + tl;
                if (q == 0 && tl < tstep && ts < p.L) p.head_y[(long long)b * p.L + ts] = hacc + p.head_b[0];
            }
        }
        lds_barrier();  
    }
    if (p.amax_out) amax_flush(omax, amax_at(p.amax_out, omax_b));
}

template <int C, bool SC, int NP = 2>
__global__ __launch_bounds__(256, (rb6_occupancy<C, SC, NP>())) void rb_fused6_kernel(const RbFused6Params p) {
    rb_fused6_body<C, SC, NP, false>(p);
}
// the 64-channel identity-shortcut block with the decoder's final conv folded in (RbFused6Params::head_*)
#ifndef RB6_HEAD_OCC
#define RB6_HEAD_OCC 2
#endif
__global__ __launch_bounds__(256, RB6_HEAD_OCC) void rb_fused6_head_kernel(const RbFused6Params p) {
    rb_fused6_body<64, false, 2, true>(p);
}

}  // namespace ac
