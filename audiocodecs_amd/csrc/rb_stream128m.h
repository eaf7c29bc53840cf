// rb_stream128m: Mimi's 128-channel identity-shortcut residual block (MimiResnetBlock at 6 kHz: 7.9 GB in + out at 128 clips x 10 s) as a
// stream kernel (rb_stream6.h: weights in LDS, one wave = one stream of 16-row tiles, accumulators are operands) -- WITHOUT a slab.
// The weight images of this block are 96 KB (k3 conv, 64 x 384) + 32 KB (1x1 conv, 128 x 64): with them in LDS there is no room for a
// per-wave copy of ELU(x) (9 KB at 128 channels), so the k3 conv's row offsets are taken in REGISTERS: a tap that reads row li - s is the
// lane's own operand fragment moved s lanes up inside its 16-lane row (v_mov_b32_dpp row_shr:s), and the s lanes that fall off the row's
// start take the previous tile's last rows from a 1 KB halo area per wave (read into the destination first: row_shr leaves the lanes
// without a source untouched).
// SC (EnCodec's block: a 1x1 shortcut conv instead of the identity, another 64 KB of weight planes): those do not fit LDS either way; the
// shortcut's fragments are read from the packed image in L2 per tile (32 KB... x 2 planes = 64 KB per 16 rows and wave through the CU's L1
// path, under the MFMAs of sixteen waves), its operand is the raw rows split in registers (rb_stream6.h); reflect padding at the clip start.
// Arithmetic, scales and accumulation order as rb_stream6.h / rb128_fused6 (k-steps in order, lo hi / hi lo / hi hi); zero padding (Mimi).
#pragma once
#include "enc_stream.h"

namespace ac {

struct RbStream128Params {
    const float* xr;         // [B][L][128] raw input
    const __bf16* w3f;       // k3 conv   [4 n-tiles][12 k-steps][2 planes][64][8]   (permuted columns: core.h perm32)
    const __bf16* wff;       // 1x1 conv  [8][2][2][64][8];  SC: [1x1 | shortcut] [8][6][2][64][8] (k-steps 0, 1 hidden, 2..5 the raw rows)
    const float *b3, *winv3; // [64]
    const float *bf, *winvf; // [128]
    float* y;                // optional raw output [B][L][128]
    float* y_elu;            // optional ELU'd output
    int B, L;
    int nseg, seg_rows;
    const unsigned* amax_in;
    unsigned* amax_out;
    float hb0, hb1;          // |hidden| <= hb0 + hb1 amax(x)
    int pad, Lp;             // SC: PAD_REFLECT and its base length (rb_fused6.h); the identity form pads with zeros
};

constexpr int R128_WA = 0, R128_WB = 98304, R128_CONST = R128_WB + 32768;               // byte offsets
constexpr int R128_B3 = 0, R128_I3 = 64, R128_BF = 128, R128_IF = 256, R128_CONST_FLOATS = 384;
constexpr int R128_HALO = 2 * 4 * 2 * 4 * 16;                                           // per wave: [plane 2][kc 4][row 2][kq 4] units of 16 B
constexpr int R128_SHARED_BYTES = R128_CONST + R128_CONST_FLOATS * 4;
template <int WAVES> constexpr size_t r128_lds() { return (size_t)R128_SHARED_BYTES + (size_t)WAVES * R128_HALO; }

template <int WAVES, bool SC, bool YR, bool YE>
__global__ __launch_bounds__(64 * WAVES) void rb_stream128m_kernel(const RbStream128Params p) {
    constexpr int KSB = SC ? 6 : 2;                              // k-steps of the second image
    constexpr int GRP = 4;                                        // output tiles finished together
    extern __shared__ __attribute__((aligned(16))) unsigned char r128_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    {
        u32x4_t* d = reinterpret_cast<u32x4_t*>(r128_smem);
        for (int i = tid; i < 98304 / 16; i += 64 * WAVES) d[R128_WA / 16 + i] = reinterpret_cast<const u32x4_t*>(p.w3f)[i];
        // the hidden k-steps of the second image: tile c's k-steps 0, 1 (4 KB) out of its KSB
        for (int i = tid; i < 32768 / 16; i += 64 * WAVES) d[R128_WB / 16 + i] = reinterpret_cast<const u32x4_t*>(p.wff)[(i >> 8) * (KSB * 128) + (i & 255)];
        float* cs = reinterpret_cast<float*>(r128_smem + R128_CONST);
        for (int e = tid; e < R128_CONST_FLOATS; e += 64 * WAVES)
            cs[e] = e < R128_I3 ? p.b3[e] : e < R128_BF ? p.winv3[e - R128_I3] : e < R128_IF ? p.bf[e - R128_BF] : p.winvf[e - R128_IF];
    }
    __syncthreads();

    const unsigned char* wa_l = r128_smem + R128_WA + lane * 16;
    const unsigned char* wb_l = r128_smem + R128_WB + lane * 16;
    // SC: the shortcut's fragments come from L2 through a buffer descriptor (lane offset in a VGPR, fragment offset in an SGPR: with flat
    // 64-bit addresses hipcc kept one address pair per fragment, 128 registers, and spilled them)
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wff, 0, SC ? 8 * 6 * 2 * 1024 : 0, 0x00020000);
    const float* c_l = reinterpret_cast<const float*>(r128_smem + R128_CONST) + 4 * kq;
    unsigned char* halo = r128_smem + R128_SHARED_BYTES + wave * R128_HALO;
    auto hunit = [&](int pl, int kc, int row) { return halo + (((pl * 4 + kc) * 2 + row) * 4 + kq) * 16; };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int clip_bytes = p.L * 512;
    const int total = p.B * p.nseg;

    for (int sg = blockIdx.x * WAVES + wave; sg < total; sg += gridDim.x * WAVES) {
        const int b = sg / p.nseg;
        const int t_beg = (sg - b * p.nseg) * p.seg_rows;
        const int t_end = t_beg + p.seg_rows < p.L ? t_beg + p.seg_rows : p.L;
        const long long ob = (long long)b * p.L * 128;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xr + ob), 0, clip_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(YR ? p.y + ob : nullptr), 0, YR ? clip_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(YE ? p.y_elu + ob : nullptr), 0, YE ? clip_bytes : 0, 0x00020000);
        const Rb16Scale cs = rb16_scale(*amax_at(p.amax_in, b), p.hb0, p.hb1);
        unsigned omax = 0;

        // rows t .. t + 15 in operand shape: r[kc][h] = channels 32 kc + 16 h + 4 kq + {0..3} of row t + li (rows outside the clip: zeros)
        auto request = [&](int t, f32x4 (&r)[4][2]) {
            const int row = t + li;
            const int ro = row >= 0 && row < p.L ? row * 512 + kq * 16 : 0x7fff0000;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                for (int h = 0; h < 2; ++h) r[kc][h] = bufload16(rs, ro + kc * 128 + h * 64, 0);
        };
        // the halo area <- rows 14, 15 of a tile's ELU(x) planes (lanes li >= 14)
        auto save_halo = [&](const Hl8 (&xe)[4]) {
            if (li >= 14) {
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    *reinterpret_cast<f16x8*>(hunit(0, kc, li - 14)) = xe[kc].hi;
                    *reinterpret_cast<f16x8*>(hunit(1, kc, li - 14)) = xe[kc].lo;
                }
            }
        };
        // fragment of the rows s above (s = 1, 2): own fragment moved s lanes up, the first s lanes from the halo (rows 2 - s + li)
        auto shifted = [&](const f16x8 own, int pl, int kc, auto s_tag) -> f16x8 {
            constexpr int S = decltype(s_tag)::value;
            const u32x4_t base = *reinterpret_cast<const u32x4_t*>(hunit(pl, kc, li < S ? 2 - S + li : 0));
            const u32x4_t o = __builtin_bit_cast(u32x4_t, own);
            u32x4_t r;
            r.x = (unsigned)__builtin_amdgcn_update_dpp((int)base.x, (int)o.x, 0x110 + S, 0xf, 0xf, false);
            r.y = (unsigned)__builtin_amdgcn_update_dpp((int)base.y, (int)o.y, 0x110 + S, 0xf, 0xf, false);
            r.z = (unsigned)__builtin_amdgcn_update_dpp((int)base.z, (int)o.z, 0x110 + S, 0xf, 0xf, false);
            r.w = (unsigned)__builtin_amdgcn_update_dpp((int)base.w, (int)o.w, 0x110 + S, 0xf, 0xf, false);
            return __builtin_bit_cast(f16x8, r);
        };

        f32x4 rx[4][2];
        {   // ---- the segment's first tile, and the two rows in front of it into the halo area (zeros left of the clip)
            f32x4 rh[4][2];
            if constexpr (SC) {                                  // rows t_beg - 16 .. t_beg - 1: lanes 14, 15 are the halo
                int j = t_beg - 16 + li;
                if (p.pad == PAD_REFLECT) j = j < 0 ? -j : (j >= p.Lp ? 2 * (p.Lp - 1) - j : j);     // ([HF]:157-176, rb_fused6.h)
                const int ho = li >= 14 && j >= 0 && j < p.L ? j * 512 + kq * 16 : 0x7fff0000;
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                    for (int h = 0; h < 2; ++h) rh[kc][h] = bufload16(rs, ho + kc * 128 + h * 64, 0);
            } else {
                request(t_beg - 16, rh);
            }
            request(t_beg, rx);
            Hl8 he[4];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) he[kc] = split16_regs8(elu4p(rh[kc][0]), elu4p(rh[kc][1]), cs.sx);
            save_halo(he);
        }

        for (int t = t_beg; t < t_end; t += 16) {
            // ---- ELU + split of the tile's rows (the raw rows stay for the identity shortcut)
            Hl8 xe[4];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) xe[kc] = split16_regs8(elu4p(rx[kc][0]), elu4p(rx[kc][1]), cs.sx);
            Hl8 xr[4];                                             // SC: the raw rows as the shortcut conv's operand (the fp32 registers are free then)
            if (SC) {
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) xr[kc] = split16_regs8(rx[kc][0], rx[kc][1], cs.sb);
            }
            // ---- stage A: hidden = ELU(conv_k3(xe) + b3); M = 64 hidden channels (4 tiles), K = 3 taps x 128
            f32x4 accA[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < 12; ++ks) {
                const int j = ks >> 2, kc = ks & 3;                // tap j reads row li + j - 2
                f16x8 xh, xl;
                if (j == 2) { xh = xe[kc].hi; xl = xe[kc].lo; }
                else if (j == 1) { xh = shifted(xe[kc].hi, 0, kc, std::integral_constant<int, 1>{}); xl = shifted(xe[kc].lo, 1, kc, std::integral_constant<int, 1>{}); }
                else { xh = shifted(xe[kc].hi, 0, kc, std::integral_constant<int, 2>{}); xl = shifted(xe[kc].lo, 1, kc, std::integral_constant<int, 2>{}); }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(wa_l + ((c * 12 + ks) * 2 + 0) * 1024);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(wa_l + ((c * 12 + ks) * 2 + 1) * 1024);
                    accA[c] = mma16(wh, wl, xh, xl, accA[c]);
                }
            }
            save_halo(xe);                                         // (behind stage A's halo reads: the wave's LDS operations stay in order)
            // the next tile's rows travel under stage B and the output (the planes' registers are free now).  SC: requested BEHIND the
            // shortcut's fragment loads -- the memory counter retires in order, and a fragment must not wait for an HBM round trip
            f32x4 rn[4][2];
            if (!SC) {
                request(t + 16 < t_end ? t + 16 : 0x3fffff00, rn);
                __builtin_amdgcn_sched_barrier(0);
            }
            Hl8 hf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 hv[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = 2 * ks + q;
                    const f32x4 b3v = *reinterpret_cast<const f32x4*>(c_l + R128_B3 + 16 * c);
                    const f32x4 i3 = *reinterpret_cast<const f32x4*>(c_l + R128_I3 + 16 * c) * cs.ix;
                    hv[q] = elu4p(es_fma4(accA[c], i3, b3v));
                }
                hf[ks] = split16_regs8(hv[0], hv[1], cs.sb);
            }
            // ---- stage B + output: y = x + W1 hidden + bf, tile c = channels 16 c + 4 kq .. = the rows' load (c >> 1, c & 1)
            const int row = t + li;
            const int orow = row < p.L ? row * 512 + kq * 16 : 0x7fff0000;
            // SC: the shortcut's fragments of output tile c (8 KB from L2) sit in ONE register set, requested while tile c - 1's values are
            // finished and stored (a second set, requested a tile earlier, measured SLOWER: 1.62 against 1.55 ms per step -- the fragments'
            // latency is not what the waves wait for); sched_barriers pin the requests where they stand
            f16x8 wsh[4], wsl[4];
            auto request_ws = [&](int c) {
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    wsh[kc] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, lane * 16, ((c * 6 + 2 + kc) * 2 + 0) * 1024, 0));
                    wsl[kc] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, lane * 16, ((c * 6 + 2 + kc) * 2 + 1) * 1024, 0));
                }
            };
            if (SC) { request_ws(0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int half = 0; half < 8 / GRP; ++half) {
                f32x4 acc[GRP];
#pragma unroll
                for (int q = 0; q < GRP; ++q) {
                    const int c = GRP * half + q;
                    f32x4 a = zero4;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const f16x8 wh = *reinterpret_cast<const f16x8*>(wb_l + ((c * 2 + ks) * 2 + 0) * 1024);
                        const f16x8 wl = *reinterpret_cast<const f16x8*>(wb_l + ((c * 2 + ks) * 2 + 1) * 1024);
                        a = mma16(wh, wl, hf[ks].hi, hf[ks].lo, a);
                    }
                    if (SC) {
#pragma unroll
                        for (int kc = 0; kc < 4; ++kc) a = mma16(wsh[kc], wsl[kc], xr[kc].hi, xr[kc].lo, a);
                        __builtin_amdgcn_sched_barrier(0);
                        if (c < 7) request_ws(c + 1);
                        else request(t + 16 < t_end ? t + 16 : 0x3fffff00, rn);   // every fragment of the tile is requested: now the next tile's rows
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const f32x4 bfv = *reinterpret_cast<const f32x4*>(c_l + R128_BF + 16 * c);
                    const f32x4 ifv = *reinterpret_cast<const f32x4*>(c_l + R128_IF + 16 * c) * cs.ib;
                    const f32x4 v = es_fma4(a, ifv, bfv);
                    if (SC) acc[q] = v;
                    else {
                        const f32x4 xv = rx[c >> 1][c & 1];
                        acc[q] = f32x4{__fadd_rn(xv.x, v.x), __fadd_rn(xv.y, v.y), __fadd_rn(xv.z, v.z), __fadd_rn(xv.w, v.w)};
                    }
                }
                {
                    unsigned tm = 0;
                    if constexpr (GRP == 4) tm = amax16(acc);
                    else { const f32x4 two[4] = {acc[0], acc[1], acc[0], acc[1]}; tm = amax16(two); }
                    omax = row < p.L && tm > omax ? tm : omax;
                }
#pragma unroll
                for (int q = 0; q < GRP; ++q) {
                    const int c = GRP * half + q;
                    if (YR) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[q]), ry, orow + c * 64, 0, 0);
                    if (YE) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, elu4p(acc[q])), re, orow + c * 64, 0, 0);
                }
            }
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                for (int h = 0; h < 2; ++h) rx[kc][h] = rn[kc][h];
        }
        if (p.amax_out) amax_flush(omax, amax_at(p.amax_out, b));
    }
}

}  // namespace ac
