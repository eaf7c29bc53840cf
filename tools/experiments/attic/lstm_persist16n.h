// lstm_persist16n: the persistent 2-layer LSTM of lstm_persist16.h -- same placement (XCD x hosts the 32 unit slices of layer x & 1 of
// clip group x >> 1), same self-validating exchange of h (fp16 hi / lo planes, bit 14 = "not yet written"), same control words, same
// tail kernel, same parameters -- with the gate columns, not K, split over the waves (round 4):
//   * FOUR waves per workgroup, one per SIMD; wave w owns units 4 w .. 4 w + 3 of the slice as 16 gate columns ordered
//     column 4 u + q (q = i, f, g, o): the four gates of a unit sit in one quad of lanes.  A wave contracts over ALL of K, so a
//     gate pre-activation is complete in its accumulator: no partial sums through LDS, no barrier between the product and the
//     gate arithmetic, and the nonlinearities run one gate per LANE (one exp + rcp per value instead of five per thread),
//     exchanged inside the quad by DPP.  lstm_persist16's step spent ~250 clocks on the partial sums and their barrier and ~1 200 on
//     the gate phase of half its waves.
//   * the wave's share of BOTH weight matrices of its layer is 256 registers -- it lives in the AccVGPRs (one wave per SIMD owns all
//     512 registers of a lane) and feeds the MFMAs from there: inline-asm MFMAs with an "a" operand (hipcc keeps the builtin's
//     operands in architectural VGPRs and copies; round 3 gave up on that).  Wait states behind the asm MFMAs are explicit (s_nop).
//     The fragments are gathered from lstm_persist16's packed image (lane (column 4 u + q, kq) <- gate tile q, lane (unit, kq)).
//   * every wave needs the whole recurrent operand (32 KB per step): the workgroup loads it ONCE, cooperatively (8 x 16 bytes per
//     thread, each piece validated by its own bit 14), into LDS in the published layout and the waves read their A fragments from
//     there; likewise the operand of the input projection (layer 0: x[t+1], split into planes on the way; layer 1: h0[t+1]).  Both
//     regions are double-buffered by step parity: ONE barrier per step.
//   * three accumulators per product (lo hi, hi lo, hi hi summed separately, then small terms first) instead of one chain of 48
//     dependent MFMAs.
// The summation order differs from lstm_persist16 (and from the per-step kernels): same parity policy, not the same bits.
#pragma once
#include "lstm_persist16.h"

namespace ac {

constexpr int LPN_XP = LP_D + 8;                       // layer 0: fp16 per row of the x planes (pitch keeps 16-byte alignment, spreads banks)
constexpr int LPN_HBYTES = LP_SLICES * LP16_SLICE_BYTES;   // 32 KB: one step's h of a 16-clip group
constexpr int LPN_PBYTES = 2 * 16 * LPN_XP * 2;        // 33 280: x[t] of 16 clips as two planes (>= LPN_HBYTES: layer 1 stages h0 there)

__device__ __forceinline__ f32x4 lpn_mfma(const f16x8 a, const f16x8 w_agpr, f32x4 c) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(w_agpr));
    return c;
}
// the accumulator may be read by vector instructions behind this (4 passes of 4 cycles + the write-back)
__device__ __forceinline__ void lpn_settle(f32x4& c) { asm volatile("s_nop 7\n\ts_nop 3" : "+v"(c)); }

template <int CTRL>
__device__ __forceinline__ float lpn_quad(float v) {     // lane j of every quad -> all four
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}

__global__ __launch_bounds__(256, 1) void lstm_persist16n_kernel(const LstmPersist16Params pp) {
    const LstmPersistParams& p = pp.base;
    constexpr int D = LP_D;
    constexpr long long GROUP_BYTES = LP16_GROUP_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char Hs[2][LPN_HBYTES];      // recurrent operand, by step parity
    __shared__ __attribute__((aligned(16))) unsigned char Ps[2][LPN_PBYTES];      // projection operand of step t: Ps[t & 1]
    __shared__ unsigned s_x, s_slot;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    unsigned* tmo = p.ctl + LP_CTL_TIMEOUT;
    if (tid == 0) {
        s_x = lp_xcc_id();
        s_slot = __hip_atomic_fetch_add(&p.ctl[LP_CTL_SLOTS + (s_x & 7) * 16], 1u, LP_RLX);
    }
    __syncthreads();
    const int x = s_x & 7, idx = s_slot;
    const int g = x >> 1, layer_rt = x & 1;
    const int G = (p.B + 15) >> 4;
    if (idx >= 32) {   // more than 32 workgroups on this XCD: the placement the roles rely on does not hold -> everybody leaves
        if (tid == 0) __hip_atomic_store(tmo, 2u, LP_RLX);
        return;
    }
    if (g >= G) return;
    if (p.dbg & 16) {   // test hook: behave like a launch whose bounded waits expired
        if (tid == 0) __hip_atomic_store(tmo, 1u, LP_RLX);
        return;
    }
    if ((p.dbg & 1) && layer_rt == 1) return;

    // (the layer as a compile-time constant: one operand path, one set of staging registers per instantiation)
    auto body = [&](auto layer_tag) {
    constexpr int layer = decltype(layer_tag)::value;
    // ---- this lane: gate q of unit eu (column 4 ul + q of the wave's 16), rows = clips 4 kq + r
    const int q = li & 3, ul = li >> 2;
    const int u0 = idx * 16, eu = u0 + 4 * wave + ul;
    // weights -> AccVGPRs: [k-step 0..15][plane], gathered from lstm_persist16's image
    //   [matrix: hh0, ih1, hh1, ih0][32 slices][4 K quarters][4 gates][4 k-steps][2 planes][64 lanes (unit, kq)][8]
    f16x8 wr[16][2], wp[16][2];                              // recurrent (W_hh of the layer), projection (W_ih of the layer)
    {
        const long long mat = (long long)LP_SLICES * 4 * 4 * 4 * 2 * 512;
        const int lane_old = kq * 16 + 4 * wave + ul;
        const __bf16* img = pp.w_pk6 + (long long)idx * (4 * 4 * 4 * 2 * 512) + lane_old * 8;
        const __bf16* pr = img + (layer == 0 ? 0 : 2) * mat;
        const __bf16* pj = img + (layer == 0 ? 3 : 1) * mat;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const int o = ((((ks >> 2) * 4 + q) * 4 + (ks & 3)) * 2 + pl) * 512;
                wr[ks][pl] = *reinterpret_cast<const f16x8*>(pr + o);
                wp[ks][pl] = *reinterpret_cast<const f16x8*>(pj + o);
            }
    }
    const float wiv = LP16_HINV * pp.winv[layer * 4 * D + q * D + eu];     // 2^-s of this lane's gate row (h travels as 2 h)
    const float bq = (layer ? p.bias1 : pp.bias0)[q * D + eu];
    const float gm = q == 2 ? 2.0f : 1.0f;                                   // tanh(x) = 2 sigmoid(2 x) - 1 for the g gate
    char* h0b = reinterpret_cast<char*>(p.hseq0);
    char* h1b = reinterpret_cast<char*>(p.hseq1);
    char* hmine = layer ? h1b : reinterpret_cast<char*>(pp.hseq0_local);
    const long long goff = (long long)(p.group0 + g) * GROUP_BYTES;

    // this lane publishes / stores row q of its four: (clip 4 kq + q, unit eu)
    const int ec = 4 * kq + q;
    const int eb = g * 16 + ec;
    const bool live = eb < p.B;
    const long long erow = (long long)(p.clip0 + eb);
    const int hpos = ec * 32 + (4 * wave + ul) * 2;           // byte offset of (clip, unit) inside a plane of the slice block

    // ---- cooperative operand loads: piece e = tid + 256 i is bytes 16 e .. of a 32 KB block
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto load_h = [&](const char* seq, int t, u32x4 (&r)[8]) {
        const char* src = seq + (long long)t * p.h_ts + goff;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, LPN_HBYTES, 0x00020000);
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + 256 * i) * 16, 0, LP_SC1);
    };
    auto h_valid = [&](const u32x4 (&r)[8]) -> bool {          // bit 14 of every fp16 clear = written
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) bad |= (r[i].x | r[i].y) | (r[i].z | r[i].w);
        return (bad & 0x40004000u) == 0u;
    };
    // wait (bounded) until this thread's eight pieces of seq[t] are written; the early request has usually brought them already
    auto h_settle = [&](const char* seq, int t, u32x4 (&r)[8]) -> bool {
        for (unsigned spins = 0;; ++spins) {
            if (__all(h_valid(r)) || (p.dbg & 4)) return true;
            if ((spins & 63) == 63 && __hip_atomic_load(tmo, LP_RLX)) return false;
            if (spins > (1u << 18)) { __hip_atomic_store(tmo, 1u, LP_RLX); return false; }
            __builtin_amdgcn_s_sleep(4);
            load_h(seq, t, r);
        }
    };
    auto put_h = [&](unsigned char* dst, const u32x4 (&r)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(dst + (tid + 256 * i) * 16) = r[i];
    };
    // layer 0: x[t] of the group's 16 clips, piece e -> clip e >> 7, floats 4 (e & 127) ..; scaled by the clip's 2^ex and split
    float xsc[8];
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.skip + (long long)(p.clip0 + g * 16) * p.skip_bs), 0, (int)(((long long)(p.B - g * 16 < 16 ? p.B - g * 16 : 16)) * p.skip_bs * 4), 0x00020000);
    if (layer == 0) {
        const int cb = p.clip0 + g * 16, cl = p.clip0 + p.B - 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = (tid >> 7) + 2 * i;
            xsc[i] = s16_pow2(s16_exponent(*amax_at(pp.amax_x, cb + c <= cl ? cb + c : cl)));
        }
    }
    auto load_x = [&](int t, f32x4 (&r)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, c = e >> 7;
            const bool ok = g * 16 + c < p.B;
            r[i] = bufload16(xrs, ok ? (int)((long long)c * p.skip_bs * 4) + (t * D + 4 * (e & 127)) * 4 : 0x7fff0000, 0);
        }
    };
    auto put_x = [&](unsigned char* dst, const f32x4 (&r)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            split16_store4s(r[i], xsc[i], dst, 16 * LPN_XP, (e >> 7) * LPN_XP + 4 * (e & 127));
        }
    };
    // rows of the projection (clips 4 kq + r) back to the recurrent product's units: * 2^(HEXP - ex) of the clip (exact)
    float xcr[4] = {1.f, 1.f, 1.f, 1.f};
    if (layer == 0) {
        const int cb = p.clip0 + g * 16, cl = p.clip0 + p.B - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) xcr[r] = s16_pow2(LP16_HEXP - s16_exponent(*amax_at(pp.amax_x, cb + kq * 4 + r <= cl ? cb + kq * 4 + r : cl)));
    }

    // ---- the two products.  A fragments from LDS: lane (clip li, kq) reads 8 k of k-step ks.  The fragments of eight k-steps are
    // requested at once, a batch ahead of the MFMAs that use them (the asm MFMAs keep their order, the scheduler would otherwise
    // sink every read next to its use: 16 exposed LDS round trips per product, 3.35 us per step).  `mid` runs between the batches.
    auto frag = [&](const unsigned char* opd, auto hl_tag, int ks, f16x8& ah, f16x8& al) {
        if constexpr (decltype(hl_tag)::value) {   // published layout: slice 2 ks + kq / 2, [plane][16 clips][16 units]
            const unsigned char* a = opd + (2 * ks + (kq >> 1)) * LP16_SLICE_BYTES + li * 32 + (kq & 1) * 16;
            ah = *reinterpret_cast<const f16x8*>(a);
            al = *reinterpret_cast<const f16x8*>(a + 512);
        } else {                                   // x planes [plane][16 clips][LPN_XP]
            const unsigned char* a = opd + (li * LPN_XP + 32 * ks + 8 * kq) * 2;
            ah = *reinterpret_cast<const f16x8*>(a);
            al = *reinterpret_cast<const f16x8*>(a + 16 * LPN_XP * 2);
        }
    };
    auto product = [&](const unsigned char* opd, auto hl_tag, const f16x8 (&w)[16][2], f32x4& out, auto&& mid) {
        f32x4 clh = {0.f, 0.f, 0.f, 0.f}, chl = clh, chh = clh;
        f16x8 h0[8], l0[8], h1[8], l1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) frag(opd, hl_tag, k, h0[k], l0[k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) frag(opd, hl_tag, 8 + k, h1[k], l1[k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            clh = lpn_mfma(l0[k], w[k][0], clh);
            chl = lpn_mfma(h0[k], w[k][1], chl);
            chh = lpn_mfma(h0[k], w[k][0], chh);
        }
        mid();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            clh = lpn_mfma(l1[k], w[8 + k][0], clh);
            chl = lpn_mfma(h1[k], w[8 + k][1], chl);
            chh = lpn_mfma(h1[k], w[8 + k][0], chh);
        }
        lpn_settle(clh); lpn_settle(chl); lpn_settle(chh);
        out = (clh + chl) + chh;
    };
    auto nothing = [] {};
    const std::integral_constant<bool, true> HL{};
    const std::integral_constant<bool, layer == 1> PL{};       // the projection operand's layout

    // ---- prologue: projection of step 0
    f32x4 accP = {0.f, 0.f, 0.f, 0.f};
    u32x4 hreq[8], preq_h[8];
    f32x4 preq_x[8];
    if (layer == 0) {
        load_x(0, preq_x);
        put_x(Ps[0], preq_x);
        load_x(p.T > 1 ? 1 : 0, preq_x);
    } else {
        load_h(h0b, 0, preq_h);
        if (!h_settle(h0b, 0, preq_h)) return;
        put_h(Ps[0], preq_h);
        load_h(h0b, p.T > 1 ? 1 : 0, preq_h);
    }
    lds_barrier();
    product(Ps[0], PL, wp, accP, nothing);
    if (layer == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) accP[r] *= xcr[r];
    }
    float cstate[4] = {0.f, 0.f, 0.f, 0.f};
    const float* skip_row = p.skip + (live ? erow : (long long)p.clip0) * p.skip_bs + eu;
    float skip_next = layer == 1 ? skip_row[0] : 0.f;
    unsigned short hist_h[LP16_BATCH], hist_l[LP16_BATCH];      // layer 0: this lane's last published terms for the batched hand-over
#pragma unroll
    for (int i = 0; i < LP16_BATCH; ++i) hist_h[i] = hist_l[i] = 0;

    for (int t = 0; t < p.T; ++t) {
        const int t1 = t + 1 < p.T ? t + 1 : t;
        // ---- operands of this step into LDS: h[t-1] (requested in the previous step's projection) and the projection operand of step
        // t + 1 (requested a step ago)
        if (t > 0) {
            if (!h_settle(hmine, t - 1, hreq)) return;
            put_h(Hs[t & 1], hreq);
        }
        if (layer == 0) put_x(Ps[t1 & 1 ? 1 : 0], preq_x);
        else {
            if (!h_settle(h0b, t1, preq_h)) return;
            put_h(Ps[t1 & 1 ? 1 : 0], preq_h);
        }
        const float skipv = skip_next;
        lds_barrier();
        // the projection operand of step t + 2: under the recurrent product and the gates, out of the exchange's way
        const int t2 = t + 2 < p.T ? t + 2 : p.T - 1;
        if (layer == 0) load_x(t2, preq_x);
        else load_h(h0b, t2, preq_h);
        // ---- pre-activations: recurrent product + the projection computed in the previous step's shadow
        f32x4 acc = accP;
        if (t > 0) {
            f32x4 rec;
            product(Hs[t & 1], HL, wr, rec, nothing);
            acc += rec;
        }
        // ---- gates: one nonlinearity per lane, the unit's four meet by DPP; every lane of the quad then holds c and h of its rows
        float hrow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pre = __fmaf_rn(acc[r], wiv, bq);
            const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-gm * pre));
            const float v = __fmaf_rn(s, gm, 1.0f - gm);
            const float ig = lpn_quad<0x00>(v), fg = lpn_quad<0x55>(v), gg = lpn_quad<0xAA>(v), og = lpn_quad<0xFF>(v);
            cstate[r] = fg * cstate[r] + ig * gg;
            hrow[r] = og * tanh_rcp(cstate[r]);
        }
        const float hn = q == 0 ? hrow[0] : q == 1 ? hrow[1] : q == 2 ? hrow[2] : hrow[3];       // this lane's (clip, unit)
        // ---- publish h[t] (no flag, no wait); a non-finite state is published as a finite stand-in and recorded (lstm_persist16.h)
        const bool nonfinite = !(fabsf(hn) < 2.0f);
        if (nonfinite && live) atomicMin(pp.poison + erow, t);
        const float h2 = LP16_HSCALE * (nonfinite ? 0.f : hn);
        _Float16 hh = (_Float16)h2;
        if (LP16_HEXP && fabsf((float)hh) >= 2.0f) hh = (_Float16)copysignf(1.9990234375f, h2);
        const _Float16 hl = (_Float16)(h2 - (float)hh);
        {
            char* dst = hmine + (long long)t * p.h_ts + goff + (long long)idx * LP16_SLICE_BYTES;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, LP16_SLICE_BYTES, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hh), rs, hpos, 0, LP_SC0);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hl), rs, 512 + hpos, 0, LP_SC0);
        }
        if (layer == 0) {   // the copy layer 1 reads from the neighbouring XCD, handed over once per batch (lstm_persist16.h)
#pragma unroll
            for (int i = 0; i < LP16_BATCH; ++i)
                if (t % LP16_BATCH == i) { hist_h[i] = __builtin_bit_cast(unsigned short, hh); hist_l[i] = __builtin_bit_cast(unsigned short, hl); }
            if ((t + 1) % LP16_BATCH == 0 || t + 1 == p.T) {
                const int tb0 = t - (t % LP16_BATCH);
#pragma unroll
                for (int i = 0; i < LP16_BATCH; ++i)
                    if (tb0 + i <= t) {
                        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(h0b + (long long)(tb0 + i) * p.h_ts + goff + (long long)idx * LP16_SLICE_BYTES), 0, LP16_SLICE_BYTES, 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b16(hist_h[i], rx, hpos, 0, LP_SC1);
                        __builtin_amdgcn_raw_buffer_store_b16(hist_l[i], rx, 512 + hpos, 0, LP_SC1);
                    }
            }
        } else if (live) {
            const float yv = hn + skipv;
            const long long o = erow * p.y_bs + (long long)t * D + eu;
            if (p.yout) p.yout[o] = yv;
            if (p.yout_elu) p.yout_elu[o] = elu1(yv);
        }
        // ---- projection of step t + 1 in the exchange's shadow; the request for the peers' h[t] goes out in its middle
        product(Ps[t1 & 1 ? 1 : 0], PL, wp, accP, [&] {
            load_h(hmine, t, hreq);
            if (layer == 1) skip_next = skip_row[(long long)t1 * D];
        });
        if (layer == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) accP[r] *= xcr[r];
        }
    }
    };
    if (layer_rt == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
}

}  // namespace ac
