// rb128_fused6: the fused residual block of rb_fused6.h for C = 128 (hidden 64), where the two weight images
// (k3 conv [64][384], [1x1 | shortcut] [128][192]) are 295 KB as bf16 planes -- they fit the register file of a CU only
// when EIGHT waves share them (144 VGPRs each, one workgroup of 512 threads per CU):
//   stage A  hidden = ELU(W3 * [xe(t-2) | xe(t-1) | xe(t)] + b3):  wave (ng = w & 3, th = w >> 2) owns hidden channels
//            16 ng .. +15 over ALL of K for HALF of the tile's time rows (round 4: with two planes per operand the full-K
//            weights of a wave are 96 registers -- rounds 1-3 split K over the wave pair and added the halves through LDS:
//            one more barrier and 17 KB of fp32 traffic per tile);
//   stage B  y = [W1 | Ws] * [hidden | x] + bf:  wave w owns output channels 16 w .. +15 and all of K.
// Everything else as in rb_fused6.h: x is read once per 64-row tile (raw rows, ELU while staging), split once into
// bf16 planes in LDS, transposed MFMA tiles (a lane holds 4 consecutive channels of one time row), 16-byte stores
// straight to HBM, the next tile is staged before the current tile's output stores are issued.  Unfused, the block
// was two tap-GEMM launches with the hidden tensor and a second read of x in HBM (1.58 ms per block at 64 x 10 s).
#pragma once
#include <hip/hip_runtime.h>
#include "rb_fused6.h"

namespace ac {

template <bool SC>
struct Rb128Cfg {
#ifndef RB128_BM
#define RB128_BM 64     // rows per tile.  96 measured no better where it fits (identity shortcut, Mimi: 6.32 vs 6.26 ms) and spills 31 registers with the 1x1 shortcut
#endif
    static constexpr int C = 128, HC = 64, BM = RB128_BM, NT = 512;
    static constexpr int KSA = 12;                                      // k-steps of stage A
    static constexpr int KSH = 2, KSB = KSH + (SC ? 4 : 0);             // k-steps of stage B: hidden, then x
    static constexpr int TT = BM / 16;                                  // 16-row time tiles (every wave covers all of them)
    static constexpr int XP = C + 8, HP = HC + 8;
    static constexpr int XE_ROWS = BM + 2;
    static constexpr int XE_PLANE = XE_ROWS * XP, XR_PLANE = SC ? BM * XP : 0, H_PLANE = BM * HP;
    static constexpr int SLOTS = (XE_ROWS * (C / 4) + NT - 1) / NT;
    static constexpr size_t lds_bytes = (size_t)2 * (XE_PLANE + XR_PLANE + H_PLANE) * 2;
    static_assert(TT % 2 == 0 && lds_bytes <= 160 * 1024, "two time halves; one workgroup per CU");
};

template <bool SC, int NP = 2>
__global__ __launch_bounds__(512, 1) void rb128_fused6_kernel(const RbFused6Params p) {
    using Cfg = Rb128Cfg<SC>;
    constexpr int C = Cfg::C, BM = Cfg::BM, NT = Cfg::NT, XP = Cfg::XP, HP = Cfg::HP, TT = Cfg::TT;
    constexpr int KSA = Cfg::KSA, KSH = Cfg::KSH, KSB = Cfg::KSB, SLOTS = Cfg::SLOTS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* Xe = reinterpret_cast<__bf16*>(smem);                  // [2][XE_ROWS][XP]
    __bf16* Xr = Xe + 2 * Cfg::XE_PLANE;                           // [2][BM][XP]       (SC only)
    __bf16* Hs = Xr + 2 * Cfg::XR_PLANE;                           // [2][BM][HP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ng = wave & 3, th = wave >> 2;
    const int li = lane & 15, kq = lane >> 4;
    const int total = p.B * p.ntiles;

    // ---- this wave's weight fragments -> registers (once)
    constexpr int WPL = NP == 2 ? 2 : 3;                           // planes in the weight images
    bf16x8 w3r[KSA][3], wfr[KSB][3];
#pragma unroll
    for (int i = 0; i < KSA; ++i)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            w3r[i][pl] = *reinterpret_cast<const bf16x8*>(p.w3f + ((((long long)ng * KSA + i) * WPL + pl) * 64 + lane) * 8);
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            wfr[ks][pl] = *reinterpret_cast<const bf16x8*>(p.wff + ((((long long)wave * KSB + ks) * WPL + pl) * 64 + lane) * 8);
    const f32x4 b3v = *reinterpret_cast<const f32x4*>(p.b3 + ng * 16 + 4 * kq);
    const f32x4 bfv = *reinterpret_cast<const f32x4*>(p.bf + wave * 16 + 4 * kq);
    f32x4 i3v = {1.f, 1.f, 1.f, 1.f}, ifv = {1.f, 1.f, 1.f, 1.f};   // split16: 2^-s of this wave's weight rows
    if (NP == 2) {
        i3v = *reinterpret_cast<const f32x4*>(p.winv3 + ng * 16 + 4 * kq);
        ifv = *reinterpret_cast<const f32x4*>(p.winvf + wave * 16 + 4 * kq);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    int s_row[SLOTS], s_q4[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int e = tid + i * NT;
        s_row[i] = e / (C / 4);
        s_q4[i] = 16 * (e % (C / 4));
    }
    const int clip_bytes = p.L * C * 4;
    f32x4 rx[SLOTS];
    auto load_tile = [&](int tile) {
        const int b = tile / p.ntiles, t0 = (tile % p.ntiles) * BM;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xr + (long long)b * p.L * C), 0, clip_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            int j = t0 - p.lpad + s_row[i];                      // left pad 2 (causal) or 1 (non-causal k3): reflect ([HF]:157-176) or zeros
            if (p.pad == PAD_REFLECT) j = j < 0 ? -j : (j >= p.Lp ? 2 * (p.Lp - 1) - j : j);
            const bool ok = s_row[i] < Cfg::XE_ROWS && j >= 0 && j < p.L;
            rx[i] = bufload16(rs, ok ? j * (C * 4) + s_q4[i] : 0x7fff0000, 0);
        }
    };
    Rb16Scale sc{1.f, 1.f, 1.f, 1.f};                           // scales of the tile that is staged in LDS (rb_fused6.h)
    auto store_tile = [&](int tile) {
        if (NP == 2) sc = rb16_scale(*amax_at(p.amax_in, tile / p.ntiles), p.hb0, p.hb1);
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int row = s_row[i], q = s_q4[i] / 16;
            if (row < Cfg::XE_ROWS) {
                split_store4<NP>(elu4(rx[i]), Xe, Cfg::XE_PLANE, row * XP + 4 * q, sc.sx);
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
        const Rb16Scale cs = sc;
        if (next < total) load_tile(next);                      // in flight during both MFMA stages
        // ---- stage A: hidden channels 16 ng .. over all of K, time tiles th * TT / 2 ..
        {
            constexpr int TH = TT / 2;
            f32x4 acc[TH];
#pragma unroll
            for (int a = 0; a < TH; ++a) acc[a] = zero4;
#pragma unroll
            for (int ks = 0; ks < KSA; ++ks) {                   // k-step of 32 over k = tap * 128 + ci
                const int j = ks >> 2, kc = ks & 3;
#pragma unroll
                for (int a = 0; a < TH; ++a) {
                    bf16x8 xf[3];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        xf[pl] = *reinterpret_cast<const bf16x8*>(Xe + pl * Cfg::XE_PLANE + ((th * TH + a) * 16 + li + j) * XP + kc * 32 + 8 * kq);
                    acc[a] = mma6<NP>(w3r[ks], xf, acc[a]);
                }
            }
#pragma unroll
            for (int a = 0; a < TH; ++a) {                       // true units: exact power-of-two factors, then the bias
                const f32x4 iv = i3v * cs.ix;
                const f32x4 v = f32x4{__fmaf_rn(acc[a].x, iv.x, b3v.x), __fmaf_rn(acc[a].y, iv.y, b3v.y), __fmaf_rn(acc[a].z, iv.z, b3v.z), __fmaf_rn(acc[a].w, iv.w, b3v.w)};
                split_store4<NP>(elu4(v), Hs, Cfg::H_PLANE, ((th * TH + a) * 16 + li) * HP + ng * 16 + 4 * kq, cs.sb);
            }
        }
        lds_barrier();  
        // identity shortcut: the x values of the output loop, requested here to arrive under stage B (rb_fused6.h)
        const int tile_b = tile / p.ntiles, tile_t0 = (tile % p.ntiles) * BM;
        f32x4 xsc[SC ? 1 : TT];
        if constexpr (!SC) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xr + (long long)tile_b * p.L * C), 0, clip_bytes, 0x00020000);
#pragma unroll
            for (int a = 0; a < TT; ++a) {
                const int t = tile_t0 + a * 16 + li;
                xsc[a] = bufload16(rs, (t < p.L ? t * (C * 4) : 0x7fff0000) + (wave * 16 + 4 * kq) * 4, 0);
            }
        }
        // ---- stage B: y = [W1 | Ws] * [hidden | x]^T + bf, output channels 16 wave ..
        f32x4 acc[TT];
#pragma unroll
        for (int a = 0; a < TT; ++a) acc[a] = NP == 2 ? zero4 : bfv;
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
#pragma unroll
            for (int a0 = 0; a0 < TT; a0 += 2) {
                bf16x8 xf[2][3];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        xf[a][pl] = ks < KSH ? *reinterpret_cast<const bf16x8*>(Hs + pl * Cfg::H_PLANE + ((a0 + a) * 16 + li) * HP + ks * 32 + 8 * kq)
                                             : *reinterpret_cast<const bf16x8*>(Xr + pl * Cfg::XR_PLANE + ((a0 + a) * 16 + li) * XP + (ks - KSH) * 32 + 8 * kq);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a0 + a] = mma6<NP>(wfr[ks], xf[a], acc[a0 + a]);
            }
        }
        lds_barrier();                                          // every wave is done reading the slabs
        if (next < total) store_tile(next);                     // staged before the output stores are issued (rb_fused6.h)
        {
            const int b = tile_b, t0 = tile_t0;
            if (p.amax_out && b != omax_b) {                     // clip change: hand the finished clip's maximum over
                amax_flush(omax, amax_at(p.amax_out, omax_b));
                omax = 0;
                omax_b = b;
            }
            const long long ob = (long long)b * p.L * C;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y ? p.y + ob : nullptr), 0, p.y ? clip_bytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_elu ? p.y_elu + ob : nullptr), 0, p.y_elu ? clip_bytes : 0, 0x00020000);
#pragma unroll
            for (int a = 0; a < TT; ++a) {
                const int t = t0 + a * 16 + li;
                const int o = (t < p.L ? t * (C * 4) : 0x7fff0000) + (wave * 16 + 4 * kq) * 4;   // rows past the clip: dropped
                f32x4 v = acc[a];
                if (NP == 2) {
                    const f32x4 iv = ifv * cs.ib;
                    v = f32x4{__fmaf_rn(v.x, iv.x, bfv.x), __fmaf_rn(v.y, iv.y, bfv.y), __fmaf_rn(v.z, iv.z, bfv.z), __fmaf_rn(v.w, iv.w, bfv.w)};
                }
                if constexpr (!SC) {                             // identity shortcut: x + block(x)
                    const f32x4 xv = xsc[a];
                    v = f32x4{__fadd_rn(xv.x, v.x), __fadd_rn(xv.y, v.y), __fadd_rn(xv.z, v.z), __fadd_rn(xv.w, v.w)};
                }
                if (p.amax_out && t < p.L) amax_acc4(omax, v);
                if (p.y) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ry, o, 0, 0);
                if (p.y_elu) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, elu4(v)), re, o, 0, 0);
            }
        }
        lds_barrier();  
    }
    if (p.amax_out) amax_flush(omax, amax_at(p.amax_out, omax_b));
}

}  // namespace ac
