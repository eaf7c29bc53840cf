// Fused SEANet residual block for the thin (32/64-channel) stages, where the unfused version is HBM-bound:
//     y = shortcut_1x1(x) + conv_1x1(ELU(conv_k3(ELU(x))))            ([HF] EncodecResnetBlock :252-282)
// One kernel reads the block input once per flavour (ELU'd rows for the k3 conv, raw rows for the
// shortcut), keeps the hidden activation (C/2 channels) in LDS and writes ELU(y) (what the next conv
// reads) -- the hidden tensor's HBM round trip and one re-read of x disappear:
//     unfused  (C=32, per time step): 128 r + 64 w | 64 r + 128 r + 128 w  = 512 B
//     fused                         : 128 r + 128 r + 128 w               = 384 B
// Persistent workgroups walk the (clip, time-tile) list; the next tile's rows are loaded into
// registers while the current tile is in the MFMA phases (v_mfma_f32_16x16x4_f32, K order as in
// tap_gemm.h: taps ascending, channels in 16-wide steps).  Weights live in REGISTERS as MFMA B fragments for
// the whole kernel (waves split the output columns NSPLIT ways so a wave's share fits), which keeps LDS at
// ~45 KB per workgroup: two to three workgroups per CU hide each other's load / barrier phases.
#pragma once
#include <hip/hip_runtime.h>
#include "tap_gemm.h"

namespace ac {

struct RbFusedParams {
    const float* xe;    // [B][L][C] ELU'd input, or null: ELU is applied to the raw rows while staging (the
                        // C=32 block is HBM-bound, so a second flavour of x in HBM costs more than the VALU work)
    const float* xr;    // [B][L][C] raw input
    const float* w3;    // packed [C/2][3C]   (k = tap*C + ci)
    const float* b3;    // [C/2]
    const float* wf;    // packed [C][C/2 + C] (k < C/2: 1x1 over the hidden; then the shortcut over x)
    const float* bf;    // [C]  (b_1x1 + b_shortcut)
    float* y;           // optional raw output [B][L][C]
    float* y_elu;       // optional ELU'd output
    int B, L, Lp;       // Lp: reflect base length (L, or 3 when L <= 2: [HF]:148-155)
    int ntiles;         // tiles per clip
    int pad;            // PAD_REFLECT (EnCodec) / PAD_ZERO (Mimi)
};

// SC: the shortcut is a 1x1 conv fused as extra K columns (EnCodec); !SC: identity shortcut (Mimi,
// [HF] mimi :421-424), added from the raw slab after the MFMA phase.
template <int C, int BM, int NSPLIT, bool SC = true>
struct RbCfg {
    static constexpr int HC = C / 2, CP = C + 4, HP = HC + 4;
    static constexpr int K3 = 3 * C, KF = SC ? HC + C : HC;
    static constexpr int XE_ROWS = BM + 2;
    static constexpr int XE_FLOATS = XE_ROWS * CP, XR_FLOATS = BM * CP, H_FLOATS = BM * HP;
    static constexpr int XE_SLOTS = (XE_ROWS * (C / 4) + 255) / 256, XR_SLOTS = (BM * (C / 4) + 255) / 256;
    static constexpr int MW = 4 / NSPLIT;                    // wave groups along time
    static constexpr int MS = BM / MW / 16;                  // 16-row sub-tiles per wave
    static constexpr int NA = HC / NSPLIT / 16, NB = C / NSPLIT / 16;   // 16-column sub-tiles per wave (stage A / B)
    static constexpr size_t lds_bytes = (size_t)(XE_FLOATS + XR_FLOATS + H_FLOATS) * 4;
};

template <int C, int BM, int NSPLIT, bool SC = true>
__global__ __launch_bounds__(256) void rb_fused_kernel(const RbFusedParams p) {
    using Cfg = RbCfg<C, BM, NSPLIT, SC>;
    constexpr int HC = Cfg::HC, CP = Cfg::CP, HP = Cfg::HP, K3 = Cfg::K3, KF = Cfg::KF;
    constexpr int MS = Cfg::MS, NA = Cfg::NA, NB = Cfg::NB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xe = smem;                                          // [XE_ROWS][CP]
    float* Xr = Xe + Cfg::XE_FLOATS;                           // [BM][CP]  (reused as the output tile)
    float* Hs = Xr + Cfg::XR_FLOATS;                           // [BM][HP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mg = wave / NSPLIT, ng = wave % NSPLIT;
    const int li = lane & 15, kq = lane >> 4;
    const int total = p.B * p.ntiles;
    const int r0 = mg * (BM / Cfg::MW);                        // this wave's first row in the tile
    const int na0 = ng * NA * 16, nb0 = ng * NB * 16;          // this wave's first output column (stage A / B)

    // ---- this wave's weight fragments -> registers (once): lane (li, kq) holds W[n0 + 16c + li][16ks + 4kq .. +3]
    f32x4 w3r[3][C / 16][NA], wfr[KF / 16][NB];
    float b3v[NA], bfv[NB];
#pragma unroll
    for (int c = 0; c < NA; ++c) {
        b3v[c] = p.b3[na0 + c * 16 + li];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int ks = 0; ks < C / 16; ++ks)
                w3r[j][ks][c] = *reinterpret_cast<const f32x4*>(p.w3 + (long long)(na0 + c * 16 + li) * K3 + j * C + ks * 16 + 4 * kq);
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        bfv[c] = p.bf[nb0 + c * 16 + li];
#pragma unroll
        for (int ks = 0; ks < KF / 16; ++ks)
            wfr[ks][c] = *reinterpret_cast<const f32x4*>(p.wf + (long long)(nb0 + c * 16 + li) * KF + ks * 16 + 4 * kq);
    }

    f32x4 re[Cfg::XE_SLOTS], rr[Cfg::XR_SLOTS];
    auto load_tile = [&](int tile) {
        const int b = tile / p.ntiles, t0 = (tile % p.ntiles) * BM;
        const float* xe = (p.xe ? p.xe : p.xr) + (long long)b * p.L * C;
        const float* xr = p.xr + (long long)b * p.L * C;
#pragma unroll
        for (int i = 0; i < Cfg::XE_SLOTS; ++i) {
            const int e = tid + i * 256;
            const int row = e / (C / 4), q = e % (C / 4);
            re[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < Cfg::XE_ROWS) {
                int j = t0 - 2 + row;                            // causal pad of 2: reflect ([HF]:157-176) or zeros
                if (p.pad == PAD_REFLECT) j = j < 0 ? -j : (j >= p.Lp ? 2 * (p.Lp - 1) - j : j);
                if (j >= 0 && j < p.L) re[i] = *reinterpret_cast<const f32x4*>(xe + (long long)j * C + 4 * q);
            }
        }
#pragma unroll
        for (int i = 0; i < Cfg::XR_SLOTS; ++i) {
            const int e = tid + i * 256;
            const int row = e / (C / 4), q = e % (C / 4);
            rr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < BM && t0 + row < p.L) rr[i] = *reinterpret_cast<const f32x4*>(xr + (long long)(t0 + row) * C + 4 * q);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < Cfg::XE_SLOTS; ++i) {
            const int e = tid + i * 256;
            const int row = e / (C / 4), q = e % (C / 4);
            if (row < Cfg::XE_ROWS) *reinterpret_cast<f32x4*>(&Xe[row * CP + 4 * q]) = p.xe ? re[i] : elu4(re[i]);
        }
#pragma unroll
        for (int i = 0; i < Cfg::XR_SLOTS; ++i) {
            const int e = tid + i * 256;
            const int row = e / (C / 4), q = e % (C / 4);
            if (row < BM) *reinterpret_cast<f32x4*>(&Xr[row * CP + 4 * q]) = rr[i];
        }
    };

    int tile = blockIdx.x;
    if (tile >= total) return;
    load_tile(tile);
    store_tile();
    __syncthreads();
    for (; tile < total; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < total) load_tile(next);                      // in flight during both MFMA stages
        // ---- stage A: hidden = ELU(conv_k3(xe) + b3) -> Hs
        {
            f32x4 acc[MS][NA];
#pragma unroll
            for (int a = 0; a < MS; ++a)
#pragma unroll
                for (int c = 0; c < NA; ++c) acc[a][c] = f32x4{b3v[c], b3v[c], b3v[c], b3v[c]};
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int ks = 0; ks < C / 16; ++ks) {
                    f32x4 af[MS];
#pragma unroll
                    for (int a = 0; a < MS; ++a) af[a] = *reinterpret_cast<const f32x4*>(&Xe[(r0 + a * 16 + li + j) * CP + ks * 16 + 4 * kq]);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int a = 0; a < MS; ++a)
#pragma unroll
                            for (int c = 0; c < NA; ++c)
                                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][u], w3r[j][ks][c][u], acc[a][c], 0, 0, 0);
                }
#pragma unroll
            for (int a = 0; a < MS; ++a)
#pragma unroll
                for (int c = 0; c < NA; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Hs[(r0 + a * 16 + kq * 4 + r) * HP + na0 + c * 16 + li] = elu1(acc[a][c][r]);
        }
        __syncthreads();
        // ---- stage B: y = [hidden | xr] * [W1; Ws] + bf
        f32x4 acc[MS][NB];
#pragma unroll
        for (int a = 0; a < MS; ++a)
#pragma unroll
            for (int c = 0; c < NB; ++c) acc[a][c] = f32x4{bfv[c], bfv[c], bfv[c], bfv[c]};
#pragma unroll
        for (int ks = 0; ks < KF / 16; ++ks) {
            f32x4 af[MS];
#pragma unroll
            for (int a = 0; a < MS; ++a)
                af[a] = ks < HC / 16 ? *reinterpret_cast<const f32x4*>(&Hs[(r0 + a * 16 + li) * HP + ks * 16 + 4 * kq])
                                     : *reinterpret_cast<const f32x4*>(&Xr[(r0 + a * 16 + li) * CP + (ks - HC / 16) * 16 + 4 * kq]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < MS; ++a)
#pragma unroll
                    for (int c = 0; c < NB; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][u], wfr[ks][c][u], acc[a][c], 0, 0, 0);
        }
        __syncthreads();                                        // every wave is done reading Xr: reuse it as the output tile
#pragma unroll
        for (int a = 0; a < MS; ++a)
#pragma unroll
            for (int c = 0; c < NB; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* o = &Xr[(r0 + a * 16 + kq * 4 + r) * CP + nb0 + c * 16 + li];
                    *o = SC ? acc[a][c][r] : __fadd_rn(*o, acc[a][c][r]);   // identity shortcut: x + block(x), element owned by this lane
                }
        __syncthreads();
        {
            const int b = tile / p.ntiles, t0 = (tile % p.ntiles) * BM;
            const long long ob = (long long)b * p.L * C;
            for (int e = tid; e < BM * (C / 4); e += 256) {
                const int row = e / (C / 4), q = e % (C / 4);
                if (t0 + row < p.L) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&Xr[row * CP + 4 * q]);
                    const long long o = ob + (long long)(t0 + row) * C + 4 * q;
                    if (p.y) *reinterpret_cast<f32x4*>(p.y + o) = v;
                    if (p.y_elu) *reinterpret_cast<f32x4*>(p.y_elu + o) = elu4(v);
                }
            }
        }
        __syncthreads();                                        // output tile drained before the slabs are refilled
        if (next < total) store_tile();
        __syncthreads();
    }
}

}  // namespace ac
