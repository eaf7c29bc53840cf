// rvq16: the codebook search of rvq.h (same residual layout, same distance form, same first-index argmax, same fp32 residual
// update) with the 2 x.E^T products in split16 arithmetic (split16.h) on the fp16 matrix pipe instead of exact fp32 products on
// v_mfma_f32_16x16x4_f32: the search was bound by the fp32 matrix pipe (0.98 ms of the 19 ms step at 0.66 of its 157 TF); three
// fp16 partial products per fp32 product run 16 / 3 times faster.
//   * residual rows: one power-of-two scale per FRAME from the row's largest magnitude, recomputed every stage (the residual
//     shrinks by stages); hi / lo planes built in registers in the order the lane already holds its values -- the k order of a
//     dot product is free as long as both operands use the same one, so the codebook image is packed in THAT order offline:
//         k-step s of 32, half e of lane (j, kq)  <->  dim 16 (2s + e/4) + 4 kq + e%4
//   * codebook stage k: one scale for the whole [C][H] table (code vectors of one table have comparable magnitudes; elements
//     16 bits below the table's largest keep fp32-grade relative precision, split16.h), image
//         epk16[K][C/16 code tiles][H/32 k-steps][2 planes][64 lanes][8 fp16]
//   * dist = -((|x|^2 - 2 dot) + |e|^2) with |x|^2 and |e|^2 exact fp32 as before; dot = acc * 2^-(sx + se) is exact scaling.
// One wave owns 16 MS frames; MS = 3 at the benchmark size: 1000 waves for 1024 SIMDs, 393 registers, and a code tile fetched
// from L2 (8 KB) feeds 36 MFMAs; tile t's distances and argmax run beside tile t + 1's MFMAs (round 6, below).
// The error of a dot product equals that of an fp32 FMA chain (split16.h); the token policy (exact outside fp64 near-ties of
// 1e-4 relative margin) holds unchanged (tests/test_gpu_parity.py, test_gpu_fullsize.py).
#pragma once
#include <hip/hip_runtime.h>
#include "rvq.h"
#include "split16.h"

namespace ac {

struct RvqEnc16Params {
    RvqEncParams base;
    const _Float16* epk16;   // split16 images of the codebooks (layout above)
    const float* einv;       // [K] 2^-se of each table's image
};

// K1: a single stage (WavTokenizer: one codebook of 4096 x 512) -- no residual update, so the 128 registers of the fp32 row are free once
// its planes are built and the code tiles can stay double-buffered at H = 512
// WS (round 4): waves per frame group -- the batch-1 regime has 5 - 47 frame groups for 1024 SIMDs and a wave walks 8 stages x 64
// code tiles at one L2 round trip per tile (0.17 ms of a 1.3 ms call): WS = 4 waves take every fourth tile each and combine their
// (best, first index) through LDS once per stage; every wave keeps the whole residual.  Same distances, same first-index rule.
template <int HV, int MS, bool CDIST, bool K1 = false, int WS = 1>
__global__ __launch_bounds__(64 * WS) void rvq_encode16_kernel(const RvqEnc16Params q) {
    static_assert(HV % 2 == 0, "k-steps of 32 dims");
    static_assert(WS == 1 || (MS == 1 && !K1), "the shared form is the small-batch form");
    constexpr int KS = HV / 2;
    const RvqEncParams& p = q.base;
    const int lane = threadIdx.x & 63;
    const int wv = WS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __shared__ float sbest[WS == 1 ? 1 : 2][WS][16];
    __shared__ int sidx[WS == 1 ? 1 : 2][WS][16];
    const int li = lane & 15, kq = lane >> 4;
    const int f0 = blockIdx.x * (16 * MS);
    const int H = p.H;

    f32x4 res[MS][HV];
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        const int frow = f0 + m * 16 + li;
#pragma unroll
        for (int v = 0; v < HV; ++v)
            res[m][v] = frow < p.F ? *reinterpret_cast<const f32x4*>(p.x + (long long)frow * p.xs + v * 16 + 4 * kq)
                                   : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ctiles = p.C / 16;
    for (int k = 0; k < (K1 ? 1 : p.K); ++k) {
        float xxr[MS][4], m2s[MS][4];        // per accumulator row: |x|^2 and -2 * 2^-(sx + se)
        f16x8 xh[MS][KS], xl[MS][KS];
        const float einv = q.einv[k];
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            float xx = 0.f;
            unsigned am = 0;
#pragma unroll
            for (int v = 0; v < HV; ++v) {
                xx = fmaf(res[m][v].x, res[m][v].x, xx); xx = fmaf(res[m][v].y, res[m][v].y, xx);
                xx = fmaf(res[m][v].z, res[m][v].z, xx); xx = fmaf(res[m][v].w, res[m][v].w, xx);
                amax_acc4(am, res[m][v]);
            }
            xx += __shfl_xor(xx, 16);
            xx += __shfl_xor(xx, 32);
            unsigned t = (unsigned)__shfl_xor((int)am, 16);
            am = t > am ? t : am;
            t = (unsigned)__shfl_xor((int)am, 32);
            am = t > am ? t : am;                              // the frame's largest finite magnitude, in all four of its lanes
            const int ex = s16_exponent(am);
            const float sx = s16_pow2(ex);
            const float mine = -2.0f * s16_pow2(-ex) * einv;   // exact: powers of two
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xxr[m][r] = __shfl(xx, kq * 4 + r);
                m2s[m][r] = __shfl(mine, kq * 4 + r);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f32x4 a = res[m][2 * s], b = res[m][2 * s + 1];
                const float v8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const _Float16 h = (_Float16)(v8[e] * sx);
                    xh[m][s][e] = h;
                    xl[m][s][e] = (_Float16)__builtin_fmaf(v8[e], sx, -(float)h);
                }
            }
        }
        float best[MS][4];
        int bidx[MS][4];
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { best[m][r] = -3.0e38f; bidx[m][r] = 0; }
        const _Float16* ep = q.epk16 + (long long)k * p.C * H * 2 + lane * 8;
        const float* eek = p.ee + (long long)k * p.C;
        auto load_tile = [&](int ct, f16x8 (&bh)[KS], f16x8 (&bl)[KS], float& eev) {
            const _Float16* et = ep + (long long)ct * KS * 1024;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bh[s] = *reinterpret_cast<const f16x8*>(et + (s * 2 + 0) * 512);
                bl[s] = *reinterpret_cast<const f16x8*>(et + (s * 2 + 1) * 512);
            }
            eev = eek[ct * 16 + li];
        };
        // a tile = its 12 MS MFMAs (mm_tile) + the distances and the running argmax (pick_tile).  The batch form below runs tile t's pick
        // BESIDE tile t + 1's MFMAs: two accumulator sets make the two independent pieces of one straight-line block, and the scheduler
        // interleaves them (~2.5 vector instructions per MFMA).  With one accumulator set a wave alternated 36 MFMAs on three dependent
        // chains with the 12 distances' compare / select chain, at one wave per SIMD: 0.396 -> 0.339 ms at 64 x 10 s, tokens identical
        // (tools/experiments/r6r_rvq_pipe.py).
        auto mm_tile = [&](const f16x8 (&bh)[KS], const f16x8 (&bl)[KS], f32x4 (&acc)[MS]) {
#pragma unroll
            for (int m = 0; m < MS; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int m = 0; m < MS; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[m][s], bh[s], acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MS; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[m][s], bl[s], acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MS; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[m][s], bh[s], acc[m], 0, 0, 0);
            }
        };
        auto pick_tile = [&](int ct, const f32x4 (&acc)[MS], float eev) {
            const int code = ct * 16 + li;
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // dist = -((xx - 2*dot) + ee) in the reference's order; -2 dot = acc * m2s is an exact scaling of the accumulator
                    float d = __fmaf_rn(acc[m][r], m2s[m][r], xxr[m][r]) + eev;
                    d = CDIST ? -sqrtf(fmaxf(d, 1e-30f)) : -d;
                    if (d > best[m][r]) { best[m][r] = d; bidx[m][r] = code; }
                }
        };
        auto run_tile = [&](int ct, const f16x8 (&bh)[KS], const f16x8 (&bl)[KS], float eev) {
            f32x4 acc[MS];
            mm_tile(bh, bl, acc);
            pick_tile(ct, acc, eev);
        };
        if constexpr (WS > 1) {            // this wave's share: tiles wv, wv + WS, ... (double-buffered like below)
            f16x8 bh0[KS], bl0[KS], bh1[KS], bl1[KS];
            float ee0, ee1 = 0.f;
            // (C % (32 WS) == 0: every wave has an even number of tiles; the look-ahead past the end re-reads the wave's last tile)
            load_tile(wv, bh0, bl0, ee0);
#ifdef RVQ16_COND_LOADS   // the first version: look-ahead loads under conditions -- run-to-run different tokens (profiles/r4_variants.md)
            for (int ct = wv; ct < ctiles; ct += 2 * WS) {
                if (ct + WS < ctiles) load_tile(ct + WS, bh1, bl1, ee1);
                run_tile(ct, bh0, bl0, ee0);
                if (ct + 2 * WS < ctiles) load_tile(ct + 2 * WS, bh0, bl0, ee0);
                if (ct + WS < ctiles) run_tile(ct + WS, bh1, bl1, ee1);
            }
#else
            for (int ct = wv; ct < ctiles; ct += 2 * WS) {
                load_tile(ct + WS, bh1, bl1, ee1);
                run_tile(ct, bh0, bl0, ee0);
                load_tile(ct + 2 * WS < ctiles ? ct + 2 * WS : ct + WS, bh0, bl0, ee0);
                run_tile(ct + WS, bh1, bl1, ee1);
            }
#endif
        } else if constexpr (HV <= 16 && MS > 1 && !K1) {   // the batch form (C % 32 == 0): two tile sets a tile ahead, two accumulator sets (above)
            f16x8 bh0[KS], bl0[KS], bh1[KS], bl1[KS];
            float ee0, ee1 = 0.f;
            f32x4 accA[MS], accB[MS];
            load_tile(0, bh0, bl0, ee0);
            load_tile(1, bh1, bl1, ee1);
            mm_tile(bh0, bl0, accA);
            for (int ct = 0; ct < ctiles; ct += 2) {
                const float e0 = ee0, e1 = ee1;
                load_tile(ct + 2 < ctiles ? ct + 2 : ct, bh0, bl0, ee0);
                mm_tile(bh1, bl1, accB);
                pick_tile(ct, accA, e0);
                load_tile(ct + 3 < ctiles ? ct + 3 : ct + 1, bh1, bl1, ee1);
                mm_tile(bh0, bl0, accA);                       // (behind the last tile: a repeat of tile ct, never picked)
                pick_tile(ct + 1, accB, e1);
            }
        } else if constexpr (HV <= 16 || K1) {    // code tiles double-buffered in registers: tile ct + 1 travels under tile ct's MFMAs
            f16x8 bh0[KS], bl0[KS], bh1[KS], bl1[KS];
            float ee0, ee1 = 0.f;
            load_tile(0, bh0, bl0, ee0);
            for (int ct = 0; ct < ctiles; ct += 2) {
                load_tile(ct + 1, bh1, bl1, ee1);
                run_tile(ct, bh0, bl0, ee0);
                if (ct + 2 < ctiles) load_tile(ct + 2, bh0, bl0, ee0);
                run_tile(ct + 1, bh1, bl1, ee1);
            }
        } else {                           // H = 512 with several stages: one tile set of 128 registers
            f16x8 bh0[KS], bl0[KS];
            float ee0;
            for (int ct = 0; ct < ctiles; ++ct) {
                load_tile(ct, bh0, bl0, ee0);
                run_tile(ct, bh0, bl0, ee0);
            }
        }
#pragma unroll
        for (int m = 0; m < MS; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int sh = 1; sh < 16; sh <<= 1) {
                    const float ob = __shfl_xor(best[m][r], sh);
                    const int oi = __shfl_xor(bidx[m][r], sh);
                    if (ob > best[m][r] || (ob == best[m][r] && oi < bidx[m][r])) { best[m][r] = ob; bidx[m][r] = oi; }
                }
            }
            if constexpr (WS > 1) {            // the waves' shares meet: larger distance value wins, equal values go to the lower index
                if (li == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sbest[k & 1][wv][kq * 4 + r] = best[m][r]; sidx[k & 1][wv][kq * 4 + r] = bidx[m][r]; }
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    best[m][r] = sbest[k & 1][0][kq * 4 + r];
                    bidx[m][r] = sidx[k & 1][0][kq * 4 + r];
#pragma unroll
                    for (int w2 = 1; w2 < WS; ++w2) {
                        const float ob = sbest[k & 1][w2][kq * 4 + r];
                        const int oi = sidx[k & 1][w2][kq * 4 + r];
                        if (ob > best[m][r] || (ob == best[m][r] && oi < bidx[m][r])) { best[m][r] = ob; bidx[m][r] = oi; }
                    }
                }
            }
            if (li == 0 && wv == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = f0 + m * 16 + kq * 4 + r;
                    if (f < p.F) p.toks[(long long)f * p.tK + p.tk0 + k] = (long long)bidx[m][r];
                }
            }
            int myidx = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int v = __shfl(bidx[m][r], (li >> 2) * 16);
                if ((li & 3) == r) myidx = v;
            }
            if (!K1 && k + 1 < p.K) {
                const float* qv_ = p.e + ((long long)k * p.C + myidx) * H + 4 * kq;
#pragma unroll
                for (int v = 0; v < HV; ++v) {
                    const f32x4 qv = *reinterpret_cast<const f32x4*>(qv_ + v * 16);
                    res[m][v].x -= qv.x; res[m][v].y -= qv.y; res[m][v].z -= qv.z; res[m][v].w -= qv.w;
                }
            }
        }
    }
}

}  // namespace ac
