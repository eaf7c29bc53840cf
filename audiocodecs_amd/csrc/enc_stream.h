// enc_stream: the thin-channel head of the EnCodec encoder as ONE kernel at sixteen waves per CU (round 6; enc_front.h is the round-3 form)
//     sig [B][T]  ->  Conv1d(1, 32, k7)  ->  ResnetBlock(32)  ->  ELU  ->  Conv1d(32, 64, k4, s2)  ->  y [B][ceil(T/2)][64]
// ([HF] modeling_encodec.py:290-301 EncodecEncoder.layers[0..3], called from audiocodecs/encodec.py:90).
// enc_front.h holds the weights of two of its three matrix stages in registers (160 VGPRs), runs two waves per SIMD and evaluates the
// stem as 112 scalar FMAs + 56 LDS reads per lane and chunk.  Here, as in rb_stream6.h:
//   * ALL weights live in LDS (50 KB of fragment images, loaded once per 16-wave workgroup); a wave needs < 128 VGPRs;
//   * one WAVE = one stream over a segment of one clip, 32 input samples per chunk, no workgroup barrier in the loop;
//   * the stem runs on the matrix pipe: x0^T = W0 [32 x (7 taps -> K 32)] * window^T, the window operand built per lane from the wave's
//     sample window in LDS (lanes kq = 0 carry the 7 taps, the others zeros), split16 like every other product;
//   * accumulators ARE operands: an accumulator tile's lane (li, kq) holds channels 16 c + 4 kq + {0..3} of row li, and the weight images
//     are packed so that an operand's element e of lane (li, kq) is channel 4 kq + e (e < 4) / 16 + 4 kq + e - 4 (core.h perm32).  x0 (raw,
//     for the shortcut) and the hidden activation go accumulator -> split -> operand registers; only the two tensors a conv reads at
//     several ROW offsets pass through the wave's slab: ELU(x0) (k3 conv) and ELU(y1) (strided conv), 16-byte units, one region reused.
// Slab per wave, per kq block of 80 units (16 B):  [0..3] XH = ELU(x0) rows 32, 33 of the previous chunk (2 planes x 2 rows);
//   [4 + 34 pl + r] the 34 rows of the tensor being read (r = 0, 1 halo; r = 2 + time - t0);  [72..75] YH = ELU(y1) rows 32, 33 of the
//   previous chunk.  kq blocks 80 units apart and rows one unit apart: the k3 conv's reads are bank-conflict free (rb_stream6.h).
// Edges ([HF]:157-176) as in enc_front.h: reflect on the left of every conv (sample window; row copies for ELU(x0) and ELU(y1)), one
// reflected step on the right of the strided conv when T is odd, length mask on the samples; a segment inside a clip runs one warm-up chunk.
// Scales: bounds derived from amax(sig) (enc_front.h header); summation order differs from enc_front.h / the separate kernels
// (fp32-faithful all the same: tests/test_fused_chains_gpu.py compares the three).
#pragma once
#include "rb_stream6.h"

namespace ac {

struct EncStreamParams {
    const float* sig;        // [B][T]
    const float* rel_len;    // optional [B]
    const __bf16* w0f;       // stem image        [2 n-tiles][1 k-step][2 planes][64][8]   (tap j in column j, columns 7..31 zero)
    const __bf16* w3f;       // k3 conv           [1][3][2][64][8]                          (permuted columns: core.h perm32)
    const __bf16* wff;       // [1x1 | shortcut]  [2][2][2][64][8]
    const __bf16* wdf;       // strided conv      [4][4][2][64][8]
    const float *b0, *winv0; // [32]
    const float *b3, *winv3; // [16]
    const float *bf, *winvf; // [32]
    const float *bd, *winvd; // [64]
    float* y;                // [B][M][64] raw
    float* dbg_x0;           // test hook: optional raw stem output [B][T][32]
    float* dbg_y1;           // test hook: optional raw block output [B][T][32]
    int B, T, M;             // M = ceil(T / 2)
    int seg_chunks, segs_per_clip;
    const unsigned* amax_sig;
    unsigned* amax_out;
    float sb0, sb1, hb0, hb1, fb0, fb1h, fb1x;   // bounds (enc_front.h)
};

constexpr int ES_WAVES = 16;
constexpr int ES_W0 = 0, ES_W3 = ES_W0 + 4096, ES_WF = ES_W3 + 6144, ES_WD = ES_WF + 8192, ES_CONST = ES_WD + 32768;   // byte offsets
constexpr int ES_B0 = 0, ES_I0 = 32, ES_B3 = 64, ES_I3 = 80, ES_BF = 96, ES_IF = 128, ES_BD = 160, ES_ID = 224, ES_CONST_FLOATS = 288;
constexpr int ES_SLAB = 4 * 80 * 16, ES_SG_FLOATS = 64;                       // per wave: the slab, then the sample window (40) + a zero area (48..55)
constexpr int ES_WAVE_BYTES = ES_SLAB + ES_SG_FLOATS * 4;
constexpr int ES_SHARED_BYTES = ES_CONST + ES_CONST_FLOATS * 4;
constexpr size_t ES_LDS = (size_t)ES_SHARED_BYTES + (size_t)ES_WAVES * ES_WAVE_BYTES;
static_assert(ES_LDS <= 160 * 1024 && ES_SHARED_BYTES % 16 == 0 && ES_WAVE_BYTES % 16 == 0, "one 16-wave workgroup per CU");

__device__ __forceinline__ f32x4 es_fma4(const f32x4 v, const f32x4 s, const f32x4 b) {
    return f32x4{__fmaf_rn(v.x, s.x, b.x), __fmaf_rn(v.y, s.y, b.y), __fmaf_rn(v.z, s.z, b.z), __fmaf_rn(v.w, s.w, b.w)};
}
__device__ __forceinline__ f32x4 es_mfma(const f16x8 a, const f16x8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int es_exp(float bound) { return s16_exponent(__float_as_uint(bound) & 0x7fffffffu); }

__global__ __launch_bounds__(64 * ES_WAVES) void enc_stream_kernel(const EncStreamParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char es_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    // ---- weights and constants -> LDS (the only workgroup-wide step)
    {
        u32x4_t* d = reinterpret_cast<u32x4_t*>(es_smem);
        for (int i = tid; i < 4096 / 16; i += 64 * ES_WAVES) d[ES_W0 / 16 + i] = reinterpret_cast<const u32x4_t*>(p.w0f)[i];
        for (int i = tid; i < 6144 / 16; i += 64 * ES_WAVES) d[ES_W3 / 16 + i] = reinterpret_cast<const u32x4_t*>(p.w3f)[i];
        for (int i = tid; i < 8192 / 16; i += 64 * ES_WAVES) d[ES_WF / 16 + i] = reinterpret_cast<const u32x4_t*>(p.wff)[i];
        for (int i = tid; i < 32768 / 16; i += 64 * ES_WAVES) d[ES_WD / 16 + i] = reinterpret_cast<const u32x4_t*>(p.wdf)[i];
        float* cs = reinterpret_cast<float*>(es_smem + ES_CONST);
        for (int e = tid; e < ES_CONST_FLOATS; e += 64 * ES_WAVES) {
            float v;
            if (e < ES_I0) v = p.b0[e - ES_B0];
            else if (e < ES_B3) v = p.winv0[e - ES_I0];
            else if (e < ES_I3) v = p.b3[e - ES_B3];
            else if (e < ES_BF) v = p.winv3[e - ES_I3];
            else if (e < ES_IF) v = p.bf[e - ES_BF];
            else if (e < ES_BD) v = p.winvf[e - ES_IF];
            else if (e < ES_ID) v = p.bd[e - ES_BD];
            else v = p.winvd[e - ES_ID];
            cs[e] = v;
        }
    }
    __syncthreads();

    const unsigned char* w0_l = es_smem + ES_W0 + lane * 16;
    const unsigned char* w3_l = es_smem + ES_W3 + lane * 16;
    const unsigned char* wf_l = es_smem + ES_WF + lane * 16;
    const unsigned char* wd_l = es_smem + ES_WD + lane * 16;
    const float* c_l = reinterpret_cast<const float*>(es_smem + ES_CONST) + 4 * kq;
    unsigned char* slab = es_smem + ES_SHARED_BYTES + wave * ES_WAVE_BYTES;
    float* sg = reinterpret_cast<float*>(slab + ES_SLAB);
    // byte address of unit (pl, row) of the main area for THIS lane's kq block, row offset `li` folded in
    unsigned char* m_l = slab + (kq * 80 + 4 + li) * 16;
    auto munit = [](int pl, int row) { return (pl * 34 + row) * 16; };
    // the halo copies move 16 units = (kq, plane, row 0 / 1): lane l < 16
    unsigned char* h_main = slab + ((lane >> 2) * 80 + 4 + ((lane >> 1) & 1) * 34 + (lane & 1)) * 16;      // main rows 0 / 1
    unsigned char* h_xh = slab + ((lane >> 2) * 80 + ((lane >> 1) & 1) * 2 + (lane & 1)) * 16;
    unsigned char* h_yh = h_xh + 72 * 16;
    auto copy16 = [](unsigned char* dst, const unsigned char* src) { *reinterpret_cast<u32x4_t*>(dst) = *reinterpret_cast<const u32x4_t*>(src); };

    // ---- this wave's stream: clip b, chunks [c_first, c_last)
    const int sid = blockIdx.x * ES_WAVES + wave;
    if (sid >= p.B * p.segs_per_clip) return;
    const int b = sid / p.segs_per_clip, seg = sid - b * p.segs_per_clip;
    const int nchunks = (p.T + 31) / 32;
    const int c_first = seg * p.seg_chunks;
    const int c_last = c_first + p.seg_chunks < nchunks ? c_first + p.seg_chunks : nchunks;
    if (c_first >= c_last) return;
    if (lane < 8) sg[48 + lane] = 0.f;                           // what the stem operand's lanes kq > 0 read

    // ---- split16 scales of this clip (bounds: enc_front.h header)
    const unsigned am_sig = *amax_at(p.amax_sig, b);
    const float a_sig = __uint_as_float(am_sig);
    const float A0 = __fmaf_rn(p.sb1, a_sig, p.sb0) * 1.0000005f;
    const float Hb = __fmaf_rn(p.hb1, A0, p.hb0) * 1.0000005f;
    const float Yb = __fmaf_rn(p.fb1h, Hb, __fmaf_rn(p.fb1x, A0, p.fb0)) * 1.000001f;
    const int es = s16_exponent(am_sig), ex = es_exp(A0), eh = es_exp(Hb), eb = eh < ex ? eh : ex, ey = es_exp(Yb);
    const float ss = s16_pow2(es), is = s16_pow2(-es);         // the samples in the stem
    const float sx = s16_pow2(ex), ix = s16_pow2(-ex);         // ELU(x0) in the k3 conv
    const float sb = s16_pow2(eb), ib = s16_pow2(-eb);         // hidden and raw x0 share stage B's accumulator
    const float sy = s16_pow2(ey), iy = s16_pow2(-ey);         // ELU(y1) in the strided conv

    const float* sigb = p.sig + (long long)b * p.T;
    float alen = 3.0e38f;
    if (p.rel_len) alen = (float)p.T * p.rel_len[b];
    auto fetch_sig = [&](int t0) -> float {                    // lane i holds sample t0 - 8 + i (reflected on the left, masked)
        const int q = t0 - 8 + lane;
        const int j = q < 0 ? -q : q;
        return (lane < 40 && j < p.T && (float)j < alen) ? sigb[j] : 0.f;
    };

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    unsigned omax = 0;
    const long long yb_off = (long long)b * p.M * 64;
    int ch = c_first > 0 ? c_first - 1 : 0;                    // a segment inside a clip warms its halos up on the chunk before
    float sv = fetch_sig(ch * 32);
    for (; ch < c_last; ++ch) {
        const int t0 = ch * 32;
        const bool emit = ch >= c_first;
        if (lane < 40) sg[lane] = sv;
        if (lane < 16) copy16(h_main, h_xh);                    // ELU(x0) halo: the previous chunk's rows 32, 33
        sv = fetch_sig(ch + 1 < c_last ? t0 + 32 : 0x3fffff00); // in flight during the chunk (no next chunk: nothing is read)

        // ---- stem on the matrix pipe: x0[a][c] = W0 tile c * window(rows 16 a ..)^T;  operand element e of lane (li, 0) = sample t - 6 + e
        f32x4 x0[2][2];
        Hl8 xr[2];
        f32x4 b0v[2], i0v[2];                                   // (read once per chunk: inside the loop every row tile would re-read them)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            b0v[c] = *reinterpret_cast<const f32x4*>(c_l + ES_B0 + 16 * c);
            i0v[c] = *reinterpret_cast<const f32x4*>(c_l + ES_I0 + 16 * c) * is;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float* wp = kq == 0 ? sg + 2 + 16 * a + li : sg + 48;
            const f32x4 s0 = {wp[0], wp[1], wp[2], wp[3]};
            const f32x4 s1 = {wp[4], wp[5], wp[6], 0.f};         // (column 7 has zero weights; a zero here keeps a later sample's inf / NaN out)
            const Hl8 so = split16_regs8(s0, s1, ss);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(w0_l + (c * 2 + 0) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(w0_l + (c * 2 + 1) * 1024);
                const f32x4 acc = mma16(wh, wl, so.hi, so.lo, zero4);
                x0[a][c] = es_fma4(acc, i0v[c], b0v[c]);
                const int t = t0 + 16 * a + li;
                if (p.dbg_x0 && emit && t < p.T) *reinterpret_cast<f32x4*>(p.dbg_x0 + ((long long)b * p.T + t) * 32 + 16 * c + 4 * kq) = x0[a][c];
            }
            const Hl8 e = split16_regs8(elu4p(x0[a][0]), elu4p(x0[a][1]), sx);
            *reinterpret_cast<f16x8*>(m_l + munit(0, 2 + 16 * a)) = e.hi;
            *reinterpret_cast<f16x8*>(m_l + munit(1, 2 + 16 * a)) = e.lo;
            xr[a] = split16_regs8(x0[a][0], x0[a][1], sb);
        }
        // reflect padding of the k3 conv at the clip start ([HF]:157-176): x0e[-1] = x0e[1], x0e[-2] = x0e[2]  (rows 1, 0 <- 3, 4)
        if (t0 == 0 && lane < 16) copy16(h_main, h_main + (4 - 2 * (lane & 1)) * 16);

        // ---- stage A: hidden = ELU(W3 * [xe(t-2) | xe(t-1) | xe(t)] + b3): 16 channels, three k-steps (one per tap)
        Hl8 hf[2];
        {
            f32x4 aH[2] = {zero4, zero4}, aL[2] = {zero4, zero4};    // hi*hi and the two cross terms on separate chains (enc_front.h)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f16x8 xh[2], xl[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    xh[a] = *reinterpret_cast<const f16x8*>(m_l + munit(0, 16 * a + j));
                    xl[a] = *reinterpret_cast<const f16x8*>(m_l + munit(1, 16 * a + j));
                }
                const f16x8 w3h = *reinterpret_cast<const f16x8*>(w3_l + (j * 2 + 0) * 1024);
                const f16x8 w3l = *reinterpret_cast<const f16x8*>(w3_l + (j * 2 + 1) * 1024);
#pragma unroll
                for (int a = 0; a < 2; ++a) aL[a] = es_mfma(w3l, xh[a], aL[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) aH[a] = es_mfma(w3h, xh[a], aH[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) aL[a] = es_mfma(w3h, xl[a], aL[a]);
            }
            const f32x4 b3v = *reinterpret_cast<const f32x4*>(c_l + ES_B3);
            const f32x4 i3v = *reinterpret_cast<const f32x4*>(c_l + ES_I3) * ix;
#pragma unroll
            for (int a = 0; a < 2; ++a) hf[a] = split16_regs8(elu4p(es_fma4(aH[a] + aL[a], i3v, b3v)), zero4, sb);   // elements 4..7: K padding
        }
        if (lane < 16) copy16(h_xh, h_main + 32 * 16);          // the next chunk's ELU(x0) halo, before ELU(y1) takes the region

        // ---- stage B: y1 = [W1 | Ws] * [hidden | x0] + bf; ELU(y1) -> rows 2..33 of the region
        {
            f32x4 acc[2][2] = {{zero4, zero4}, {zero4, zero4}};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 wh[2], wl[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    wh[c] = *reinterpret_cast<const f16x8*>(wf_l + ((c * 2 + ks) * 2 + 0) * 1024);
                    wl[c] = *reinterpret_cast<const f16x8*>(wf_l + ((c * 2 + ks) * 2 + 1) * 1024);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = es_mfma(wl[c], ks ? xr[a].hi : hf[a].hi, acc[a][c]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = es_mfma(wh[c], ks ? xr[a].lo : hf[a].lo, acc[a][c]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = es_mfma(wh[c], ks ? xr[a].hi : hf[a].hi, acc[a][c]);
            }
            f32x4 bfv[2], ifv[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                bfv[c] = *reinterpret_cast<const f32x4*>(c_l + ES_BF + 16 * c);
                ifv[c] = *reinterpret_cast<const f32x4*>(c_l + ES_IF + 16 * c) * ib;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                f32x4 v[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    v[c] = es_fma4(acc[a][c], ifv[c], bfv[c]);
                    const int t = t0 + 16 * a + li;
                    if (p.dbg_y1 && emit && t < p.T) *reinterpret_cast<f32x4*>(p.dbg_y1 + ((long long)b * p.T + t) * 32 + 16 * c + 4 * kq) = v[c];
                }
                const Hl8 e = split16_regs8(elu4p(v[0]), elu4p(v[1]), sy);
                *reinterpret_cast<f16x8*>(m_l + munit(0, 2 + 16 * a)) = e.hi;
                *reinterpret_cast<f16x8*>(m_l + munit(1, 2 + 16 * a)) = e.lo;
            }
        }
        if (lane < 16) copy16(h_main, h_yh);                    // ELU(y1) halo: the previous chunk's rows 32, 33
        // clip edges of the strided conv's input ([HF]:157-176): y1e[-1] = y1e[1], y1e[-2] = y1e[2]; T odd: y1e[T] = y1e[T-2]
        if (t0 == 0 && lane < 16) copy16(h_main, h_main + (4 - 2 * (lane & 1)) * 16);
        if ((p.T & 1) && p.T < t0 + 32 && lane < 8) {
            unsigned char* e0 = slab + ((lane >> 1) * 80 + 4 + (lane & 1) * 34 + (p.T - t0)) * 16;      // row of time T - 2
            copy16(e0 + 2 * 16, e0);
        }

        // ---- stage C: y[m] = Wd * [y1e[2m-2] | y1e[2m-1] | y1e[2m] | y1e[2m+1]] + bd: 16 output rows x 64 channels
        {
            f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 xh = *reinterpret_cast<const f16x8*>(m_l + li * 16 + munit(0, ks));       // row 2 li + ks
                const f16x8 xl = *reinterpret_cast<const f16x8*>(m_l + li * 16 + munit(1, ks));
                f16x8 wh[4], wl[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    wh[c] = *reinterpret_cast<const f16x8*>(wd_l + ((c * 4 + ks) * 2 + 0) * 1024);
                    wl[c] = *reinterpret_cast<const f16x8*>(wd_l + ((c * 4 + ks) * 2 + 1) * 1024);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = es_mfma(wl[c], xh, acc[c]);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = es_mfma(wh[c], xl, acc[c]);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = es_mfma(wh[c], xh, acc[c]);
            }
            if (lane < 16) copy16(h_yh, h_main + 32 * 16);      // the next chunk's ELU(y1) halo
            const int m = (t0 >> 1) + li;
            const bool st = emit && m < p.M;
            float* yr = p.y + yb_off + (long long)m * 64 + 4 * kq;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 bdv = *reinterpret_cast<const f32x4*>(c_l + ES_BD + 16 * c);
                const f32x4 idv = *reinterpret_cast<const f32x4*>(c_l + ES_ID + 16 * c) * iy;
                acc[c] = es_fma4(acc[c], idv, bdv);
                if (st) *reinterpret_cast<f32x4*>(yr + 16 * c) = acc[c];
            }
            const unsigned cmax = amax16(acc);
            omax = st && cmax > omax ? cmax : omax;
        }
    }
    if (p.amax_out) amax_flush(omax, amax_at(p.amax_out, b));
}

}  // namespace ac
