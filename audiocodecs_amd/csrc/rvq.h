// Residual vector quantiser of [HF] EncodecResidualVectorQuantizer (:424-447) over
// EncodecEuclideanCodebook.quantize (:364-369):
//     dist = -(sum(x^2) - 2 x.E^T + sum(E^2));  idx = argmax(dist) (first index on ties);
//     residual -= E[idx];   K stages.
// Encode: one wave owns 16 frames for ALL K stages.  The residual tile lives in registers in MFMA
// A-fragment order (H/16 16-byte vectors per lane); each stage streams the codebook (pre-packed in
// B-fragment order, L2-resident: 512 KB/stage) through v_mfma_f32_16x16x4_f32, keeps a running
// (max, first index) per accumulator row, finishes the row argmax with a 16-lane butterfly
// (wavefront reduction), gathers the winning code vectors and subtracts them in registers.
// The distance matrix ([48000,1024] fp32 = 197 MB per stage in the reference) never exists.
// Decode: sum_k E_k[tok] accumulated in stage order from 0.0, one thread per 4 latent dims.
#pragma once
#include <hip/hip_runtime.h>
#include "tap_gemm.h"
#include "rvq_types.h"

namespace ac {


// CDIST = false: EnCodec's expanded form above.  CDIST = true: Mimi, [HF] mimi :985-990
//     idx = argmin(cdist(x, E, p=2))  with cdist's matmul form  sqrt(clamp_min(|x|^2 - 2 x.E^T + |E|^2, 1e-30)):
// the square root is kept because it merges near-equal squared distances into exact ties (first index wins).
template <int HV, int MS, bool CDIST>  // HV = H/16: 16-byte vectors of a residual row per lane; MS: 16-frame sub-tiles per wave
__global__ __launch_bounds__(64) void rvq_encode_kernel(const RvqEncParams p) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, kq = lane >> 4;
    const int f0 = blockIdx.x * (16 * MS);
    const int H = p.H;

    // Every code tile fetched from L2 (8 KB at H = 128) feeds MS x 32 MFMAs: with one sub-tile per wave
    // the kernel was bound by the CU's 64 B/clk L1 path (70 TF), not by the matrix pipe.
    f32x4 res[MS][HV];
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        const int frow = f0 + m * 16 + li;                 // the frame whose A-fragment this lane holds
#pragma unroll
        for (int v = 0; v < HV; ++v)
            res[m][v] = frow < p.F ? *reinterpret_cast<const f32x4*>(p.x + (long long)frow * p.xs + v * 16 + 4 * kq)
                                   : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ctiles = p.C / 16;
    for (int k = 0; k < p.K; ++k) {
        float xxr[MS][4];
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            // ||x||^2 of frame li: partial over this lane's k's, then across the 4 kq lanes
            float xx = 0.f;
#pragma unroll
            for (int v = 0; v < HV; ++v) {
                xx = fmaf(res[m][v].x, res[m][v].x, xx); xx = fmaf(res[m][v].y, res[m][v].y, xx);
                xx = fmaf(res[m][v].z, res[m][v].z, xx); xx = fmaf(res[m][v].w, res[m][v].w, xx);
            }
            xx += __shfl_xor(xx, 16);
            xx += __shfl_xor(xx, 32);
            // accumulator row r of this lane is frame kq*4 + r: fetch its ||x||^2
#pragma unroll
            for (int r = 0; r < 4; ++r) xxr[m][r] = __shfl(xx, kq * 4 + r);
        }
        float best[MS][4];
        int bidx[MS][4];
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { best[m][r] = -3.0e38f; bidx[m][r] = 0; }
        const float* ep = p.epk + (long long)k * p.C * H + lane * 4;
        const float* eek = p.ee + (long long)k * p.C;
        // code tiles are double-buffered in registers: tile ct+1 is in flight while tile ct runs its MFMAs
        auto load_tile = [&](int ct, f32x4 (&bw)[HV], float& eev) {
            const float* et = ep + (long long)ct * HV * 256;
#pragma unroll
            for (int v = 0; v < HV; ++v) bw[v] = *reinterpret_cast<const f32x4*>(et + v * 256);
            eev = eek[ct * 16 + li];
        };
        auto run_tile = [&](int ct, const f32x4 (&bw)[HV], float eev) {
            f32x4 acc[MS];
#pragma unroll
            for (int m = 0; m < MS; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < HV; ++v)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int m = 0; m < MS; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(res[m][v][u], bw[v][u], acc[m], 0, 0, 0);
            const int code = ct * 16 + li;
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // dist = -((xx - 2*dot) + ee), evaluated in the reference's order
                    float d = (xxr[m][r] - 2.0f * acc[m][r]) + eev;
                    d = CDIST ? -sqrtf(fmaxf(d, 1e-30f)) : -d;
                    if (d > best[m][r]) { best[m][r] = d; bidx[m][r] = code; }
                }
        };
        f32x4 bw0[HV], bw1[HV];
        float ee0, ee1 = 0.f;
        load_tile(0, bw0, ee0);
        for (int ct = 0; ct < ctiles; ct += 2) {           // C is a multiple of 32 in every configured codec
            load_tile(ct + 1, bw1, ee1);
            run_tile(ct, bw0, ee0);
            if (ct + 2 < ctiles) load_tile(ct + 2, bw0, ee0);
            run_tile(ct + 1, bw1, ee1);
        }
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            // row argmax across the 16 lanes holding the row's columns; ties -> smaller index
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int sh = 1; sh < 16; sh <<= 1) {
                    const float ob = __shfl_xor(best[m][r], sh);
                    const int oi = __shfl_xor(bidx[m][r], sh);
                    if (ob > best[m][r] || (ob == best[m][r] && oi < bidx[m][r])) { best[m][r] = ob; bidx[m][r] = oi; }
                }
            }
            // token out: lane li == 0 of each row group writes rows kq*4 + r
            if (li == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = f0 + m * 16 + kq * 4 + r;
                    if (f < p.F) p.toks[(long long)f * p.tK + p.tk0 + k] = (long long)bidx[m][r];
                }
            }
            // residual update: this lane needs the index of frame li = row (li>>2)*4 + (li&3)
            int myidx = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int v = __shfl(bidx[m][r], (li >> 2) * 16);
                if ((li & 3) == r) myidx = v;
            }
            if (k + 1 < p.K) {
                const float* q = p.e + ((long long)k * p.C + myidx) * H + 4 * kq;
#pragma unroll
                for (int v = 0; v < HV; ++v) {
                    const f32x4 qv = *reinterpret_cast<const f32x4*>(q + v * 16);
                    res[m][v].x -= qv.x; res[m][v].y -= qv.y; res[m][v].z -= qv.z; res[m][v].w -= qv.w;
                }
            }
        }
    }
}


__global__ __launch_bounds__(256) void rvq_decode_kernel(const RvqDecParams p) {
    const int hv = p.H / 4;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)p.F * hv) return;
    const long long f = gid / hv;
    const int q = (int)(gid % hv);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.K; ++k) {
        long long idx = p.toks[f * p.tK + p.tk0 + k];
        if (idx < 0 || idx >= p.C) {   // F.embedding raises here.  No entry point synchronises, so: the frame becomes NaN and a
            // sticky word makes the next call on the handle return AC_EINVAL
            if (q == 0 && p.bad) __hip_atomic_fetch_add(p.bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            acc.x = acc.y = acc.z = acc.w = __uint_as_float(0x7fc00000u);
            idx = 0;
        }
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.e + ((long long)k * p.C + idx) * p.H + 4 * q);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<f32x4*>(p.out + f * p.os + 4 * q) = acc;
}

}  // namespace ac
