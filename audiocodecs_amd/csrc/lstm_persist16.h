// lstm_persist16: the persistent 2-layer LSTM (lstm_persist.h: mathematics, control words, bounded waits, tail kernel)
// with a self-validating exchange of h (below), in split16.h arithmetic -- two fp16 planes per operand, 3 partial products --
// re-scheduled around what the round-2 traces of the 4-wave kernel showed (tools/ubench/xcd_exchange2.hip, AC_LSTM_DBG=32):
//   * EIGHT waves per workgroup (one workgroup per CU, two waves per SIMD): a wave owns an EIGHTH of K for the workgroup's 64 gate
//     columns, so its share of the two weight matrices of its layer is 128 registers and everything fits the 256 architectural
//     VGPRs.  With four waves the weights spilled into AccVGPRs and every MFMA was preceded by four v_accvgpr_read copies and a
//     hazard nop: the 48 MFMAs of a product took 0.6 us instead of 0.4.
//   * PLACEMENT: XCD x hosts all 32 unit slices of layer x & 1 of clip group x >> 1.  The recurrent exchange of a layer stays
//     inside one XCD's L2: h is published with workgroup-scope (sc0) stores -- complete when the L2 has them, no write-through
//     -- and read by the 32 peers with agent-scope (sc1) loads that miss the per-CU L1 and hit that L2: 1.1 us per
//     publish / read-32 KB round against 2.2 us for a role spread over an XCD pair or published with sc1 stores.
//   * layer 0 -> layer 1 crosses to the neighbouring XCD.  An sc1 store is acknowledged only when written through, and the
//     memory counter is in order, so the next recurrent load would wait for it: layer 0 keeps its last LP16_BATCH published
//     values in LDS and writes them to hseq0 once per batch; layer 1 runs that far behind.
//   * the CU's load path RETURNS IN ORDER and moves 64 B/clk: layer 1's cross-XCD requests (h0[t+2] for its projection, the skip
//     value) are issued right BEHIND the request for the next recurrent operand and have a whole step to arrive; layer 0's x[t+2]
//     (32 KB per CU and step, as much as the exchange) is requested in FRONT of the barrier and passes under the partial sums and
//     the gate arithmetic (round 4: behind the recurrent request it shared the path with the peers' h and layer 0 bound the
//     kernel).  The recurrent request itself is issued in the projection that follows the publish (the peers publish at
//     about the same time), so its round trip runs under the projection's MFMAs -- by the gate waves (0 .. 3) at the start of their
//     projection, by waves 4 .. 7, whose projection runs BESIDE the gates, when the gate waves have posted their publish (round 6:
//     LATE below); when it comes back incomplete, the step starts with the ordinary polling load.
// h lies in [-1, 1]: it travels as 2 h = hi + lo with |hi| < 2 (bit 14 clear: the exchange's "has arrived" test).  1.0 itself occurs
// when the gates saturate, and fp16(2.0) has bit 14 set: hi is capped at the largest fp16 below 2, lo takes the rest; the rows of
// [W_ih | W_hh] of a layer share one power-of-two scale per gate row; the fused layer-0 projection scales x[t] by its clip's amax
// scale and rescales the result to the recurrent product's units (exact).  A slice block is [2 planes][16 clips][16 units] fp16.
// (Rounds 1-3 carried a 4-wave ancestor with three bf16 planes per operand, lstm_persist6.h; removed in round 4 with that arithmetic.)
#pragma once
#include "lstm_persist.h"
#include "tap_gemm6.h"
#include "rb_fused6.h"
#include <type_traits>

namespace ac {

// A slice's block of h for one time step and 16-clip group: [2 planes][16 clips][16 units] fp16 (1024 B), so a consumer lane
// (clip i, kq) fetches the 8 units 32 ks + 8 kq .. of a k-step with one 16-byte load per plane from slice 2 ks + kq / 2.
// SELF-VALIDATING exchange: the host fills the h buffers with 0xFF bytes before the launch; every published fp16 term has
// |v| < 2, i.e. bit 14 (the exponent's top bit) clear, the fill pattern has it set.  A consumer therefore needs no flag: it
// loads the operand and looks at bit 14 of every element -- the data says itself whether it has arrived, in whatever order the
// memory system performs the loads.  One round trip instead of flag poll + dependent load; spins are bounded by the timeout word.
constexpr int LP16_SLICE_BYTES = 2 * 16 * 16 * 2;    // 1024
constexpr long long LP16_GROUP_BYTES = (long long)LP_SLICES * LP16_SLICE_BYTES;

struct LstmPersist16Params {
    LstmPersistParams base;     // hseq0 / hseq1 are byte buffers of fp16 plane blocks here; h_ts = bytes per time step
    const __bf16* w_pk6;        // [hh0, ih1, hh1, ih0][32 slices][4 waves][4 gates][4 k-steps of 32][2 planes][64 lanes][8 fp16] register images
    const float* bias0;         // fuse_in: b_ih0 + b_hh0 [4D]
    int fuse_in;                // 1: the layer-0 slices compute W_ih0 * x[t] themselves (x = base.skip, fp32 [B][T][D]) instead of
                                //    reading a pre-computed gin0 -- the [T*B][4D] projection GEMM and its HBM round trip disappear;
                                //    layer 0 has the slack (its step is shorter than layer 1's, which bounds the kernel)
    int* poison;                // [all clips] first time step at which a clip's state went non-finite (INT_MAX-like fill
                                //    = never): see the publish step and lstm_tail_kernel
    // per gate row 2^-s of layer 0's [W_ih0 | W_hh0] and of layer 1's [W_ih1 | W_hh1] rows ([2][4D]: the two matrices of a layer
    // share the accumulator, so their rows share the scale); amax slot of x (indexed by clip, fuse_in only)
    const float* winv;
    const unsigned* amax_x;
    void* hseq0_local;          // layer 0's own copy of h0 (same layout as hseq0)
};

#ifndef LP16_HEXP
#define LP16_HEXP 1                                  // h travels as 2^LP16_HEXP h (0: unscaled)
#endif
constexpr float LP16_HSCALE = LP16_HEXP ? 2.0f : 1.0f, LP16_HINV = LP16_HEXP ? 0.5f : 1.0f;
#ifndef LP16_BATCH_N
#define LP16_BATCH_N 4      // steps of h0 per hand-over to layer 1 (measured: 1: 4.49 ms, 2: 4.39, 4: 4.37, 8: 4.39, 16: 4.43, 32: 4.48 per EnCodec step)
#endif
constexpr int LP16_BATCH = LP16_BATCH_N;                        // steps of h0 per hand-over to layer 1
// Waves 4 .. 7 (LATE, below) request the recurrent operand when this workgroup's gate waves have issued their publish stores (an LDS
// word per gate wave) + LP16_LATE_SLEEP x 64 clocks.  Measured per EnCodec step (two launches; one box, alternating processes): request in
// the middle of the projection (rounds 4 - 5) 3.75 ms; at its end + a fixed sleep of 0 / 6 / 12 / 18 / 24 units 3.65 / 3.56 / 3.51 / 3.58 /
// 3.90; on the flag + 0 / 2 / 4 units 3.50 / 3.54 / 3.58; on a flag posted BEFORE the stores 3.68.
#ifndef LP16_PART_PITCH
#define LP16_PART_PITCH 16  // 16-byte units per clip row of the partial sums (measured per EnCodec step: 16: 3.35 ms, 17 / 18: 3.37 - 3.38; one float per access: 3.46)
#endif
#ifndef LP16_LATE_FLAG
#define LP16_LATE_FLAG 1
#endif
#ifndef LP16_LATE_SLEEP
#define LP16_LATE_SLEEP 0
#endif

// FUSE: compile-time copy of LstmPersist16Params::fuse_in (a run-time branch around the gin loads would make the compiler wait
// for ALL outstanding memory operations in front of the gate arithmetic)
template <bool FUSE>
__global__ __launch_bounds__(512) void lstm_persist16_kernel(const LstmPersist16Params pp) {
    const LstmPersistParams& p = pp.base;
    constexpr int D = LP_D, NP = 2;
    constexpr int SLICE_BYTES = LP16_SLICE_BYTES;
    constexpr long long GROUP_BYTES = LP16_GROUP_BYTES;
    // partial sums of the eight K eighths: [step parity][wave][clip][unit (+ pad)] x the FOUR gates of the unit as one 16-byte word -- a lane's
    // accumulators hold gate n of (clip kq * 4 + r, unit li) in acc[n][r], so a lane writes 4 x 16 bytes and a gate thread reads 8 x 16 bytes
    // (round 6; one float per access before: 16 writes, 32 reads on the step's critical path)
    __shared__ f32x4 part[2][8][16][LP16_PART_PITCH];
    __shared__ unsigned s_x, s_slot;
    __shared__ __attribute__((aligned(16))) unsigned pubflag[4];      // LATE: step + 1 of the last publish of each gate wave
    __shared__ unsigned short hist[LP16_BATCH][2][256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform for the compiler too (buffer descriptors depend on it)
    const int li = lane & 15, kq = lane >> 4;
    unsigned* tmo = p.ctl + LP_CTL_TIMEOUT;
    if (tid < 4) pubflag[tid] = 0;
    if (tid == 0) {
        s_x = lp_xcc_id();
        s_slot = __hip_atomic_fetch_add(&p.ctl[LP_CTL_SLOTS + (s_x & 7) * 16], 1u, LP_RLX);
    }
    __syncthreads();
    const int x = s_x & 7, slot = s_slot;
    const int g = x >> 1;
    const int G = (p.B + 15) >> 4;
    if (slot >= 32) {   // more than 32 workgroups on this XCD: the placement the roles rely on does not hold -> everybody leaves
        if (tid == 0) __hip_atomic_store(tmo, 2u, LP_RLX);
        return;
    }
    if (g >= G) return;
    if (AC_DEV_MODE(p.dbg, 16)) {   // test hook (AC_LSTM_DBG=16): behave like a launch whose bounded waits expired
        if (tid == 0) __hip_atomic_store(tmo, 1u, LP_RLX);
        return;
    }
    if (AC_DEV_MODE(p.dbg, 1) && (x & 1) == 1) return;
    const int layer_rt = x & 1, idx = slot, u0 = idx * 16;
    auto body = [&](auto layer_tag, auto late_tag) {
    constexpr int layer = decltype(layer_tag)::value;
    // LATE: waves 4 .. 7 hold no gate threads: behind the barrier they go straight into the next step's projection, ~0.5 us BEFORE waves
    // 0 .. 3 (here and on the 31 peers) have published h[t].  A recurrent request in the middle of that projection always came back
    // incomplete, the next step began with a fresh polling load for half of the K eighths, and the gate waves waited for them at the
    // barrier (0.16 us of the 2.24 us step, tools/experiments/r6u_lstm_trace.py).  These waves request at the END of their projection,
    // behind a short sleep; the gate waves, whose projection follows their own publish, at its START.  Compile-time copies of the body:
    // a request under a run-time condition becomes a register copy at the join, and the copy waits for the load (project0).
    constexpr bool LATE = decltype(late_tag)::value;
    // LATE waves: until this workgroup's four gate waves have issued the publish stores of step t (the peers do so at about the same
    // time), then LP16_LATE_SLEEP x 64 clocks for the stores to reach the L2.  Bounded: a gate wave that left on a timeout never posts.
    auto await_publish = [&](int t) {
#if LP16_LATE_FLAG
        for (int spins = 0; spins < (1 << 12); ++spins) {
            const u32x4_t f = *reinterpret_cast<volatile u32x4_t*>(pubflag);
            const unsigned lo = f.x < f.y ? f.x : f.y, lo2 = f.z < f.w ? f.z : f.w;
            if ((lo < lo2 ? lo : lo2) > (unsigned)t) break;
            if ((spins & 63) == 63 && __hip_atomic_load(tmo, LP_RLX)) break;      // the launch is being abandoned
            __builtin_amdgcn_s_sleep(1);
        }
#else
        (void)t;
#endif
        __builtin_amdgcn_s_sleep(LP16_LATE_SLEEP);
    };

    // ---- weights -> registers: [gate][k-step of 32 inside this wave's 64 k][plane].  The packed image is
    // [matrix][32 slices][4 K quarters][4 gates][4 k-steps][2 planes][64 lanes][8]: wave w = quarter w / 2, k-steps 2 (w & 1) ..
    bf16x8 wa[4][2][2], wb[4][2][2];                          // layer 0: wa = W_hh0, wb = W_ih0;  layer 1: wa = W_ih1, wb = W_hh1
    {
        const long long mat = (long long)LP_SLICES * 4 * 4 * 4 * NP * 512;     // 16-bit elements per matrix
        const __bf16* base = pp.w_pk6 + ((long long)idx * 4 + (wave >> 1)) * (4 * 4 * NP * 512) + lane * 8;
        const __bf16* pa = base + (layer == 0 ? 0 : mat);
        const __bf16* pb = base + (layer == 0 ? 3 : 2) * mat;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    const int o = ((n * 4 + (wave & 1) * 2 + ks) * NP + pl) * 512;
                    wa[n][ks][pl] = *reinterpret_cast<const bf16x8*>(pa + o);
                    if (layer || FUSE) wb[n][ks][pl] = *reinterpret_cast<const bf16x8*>(pb + o);
                    else wb[n][ks][pl] = wa[n][ks][pl];
                }
    }
    char* h0b = reinterpret_cast<char*>(p.hseq0);
    char* h1b = reinterpret_cast<char*>(p.hseq1);
    char* hmine = layer ? h1b : reinterpret_cast<char*>(pp.hseq0_local);
    const long long goff = (long long)(p.group0 + g) * GROUP_BYTES;

    // gate threads: the first 256 threads own (clip ec, unit ej) of the slice, as in the 4-wave kernels
    const bool gate_thr = tid < 256;
    const int ec = (tid & 255) >> 4, ej = tid & 15;
    const int eb = g * 16 + ec;
    const bool live = gate_thr && eb < p.B;
    const long long erow = (long long)(p.clip0 + eb);
    const int eu = u0 + ej;
    float cstate = 0.f;
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool fuse0 = layer == 0 && FUSE;
    if (layer == 1 || fuse0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = (layer ? p.bias1 : pp.bias0)[q * D + eu];
    }
    float wiv[4];                                             // 2^-s of this thread's gate rows, (h travels as 2 h)
#pragma unroll
    for (int q = 0; q < 4; ++q) wiv[q] = LP16_HINV * pp.winv[layer * 4 * D + q * D + eu];
    const int hpos = ec * 32 + ej * 2;                         // byte offset of (clip, unit) inside a plane of the slice block

    // A operand: for k-step ks of this wave's K eighth (slices 4 wave .. 4 wave + 3), lane (clip li, kq) reads units 8 kq .. 8 kq + 7
    auto load_a = [&](const char* seq, int t, bf16x8 (&a)[2][2]) {
        const char* src = seq + (long long)t * p.h_ts + goff + (long long)(wave * 4) * SLICE_BYTES;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4 * SLICE_BYTES, 0x00020000);
        const int lo = (kq >> 1) * SLICE_BYTES + li * 32 + (kq & 1) * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                a[ks][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, ks * 2 * SLICE_BYTES + pl * 512 + lo, 0, LP_SC1));
    };
    auto mac = [&](const bf16x8 (&a)[2][2], const bf16x8 (&w)[4][2][2], f32x4 (&acc)[4], int ks0 = 0, int ks1 = 2) {
#pragma unroll
        for (int ks = ks0; ks < ks1; ++ks)
#pragma unroll
            for (int n = 0; n < 4; ++n) {   // split16.h: lo hi, hi lo, hi hi on the fp16 pipe
                f32x4 v = acc[n];
                v = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[ks][1]), __builtin_bit_cast(f16x8, w[n][ks][0]), v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[ks][0]), __builtin_bit_cast(f16x8, w[n][ks][1]), v, 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[ks][0]), __builtin_bit_cast(f16x8, w[n][ks][0]), v, 0, 0, 0);
            }
    };
    auto valid = [&](const bf16x8 (&a)[2][2]) -> bool {        // bit 14 of every element clear = every piece has arrived
        unsigned bad = 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const u32x4_t w = __builtin_bit_cast(u32x4_t, a[ks][pl]);
                bad |= (w.x | w.y) | (w.z | w.w);
            }
        return __all((bad & 0x40004000u) == 0u);
    };
    // SELF-VALIDATING exchange (above): the h buffers start as 0xFF bytes; spins are bounded by the timeout word
    auto load_valid = [&](const char* seq, int t, bf16x8 (&a)[2][2]) -> bool {
        for (unsigned spins = 0;; ++spins) {
            load_a(seq, t, a);
            if (valid(a) || AC_DEV_MODE(p.dbg, 4)) return true;
            if ((spins & 63) == 63 && __hip_atomic_load(tmo, LP_RLX)) return false;
            if (spins > (1u << 18)) { __hip_atomic_store(tmo, 1u, LP_RLX); return false; }
            __builtin_amdgcn_s_sleep(4);
        }
    };
    bf16x8 arec[2][2];               // recurrent operand requested ahead (see the header)
    float skip_next = 0.f;           // layer 1: x[t+1] of this thread's (clip, unit), requested a step ahead
    f32x4 accP[4];
    // layer 0 (fused input projection): x[t] of clip li, this wave's K eighth: 16 fp32 per lane in registers, requested one step
    // ahead right behind the recurrent request (plain loads: the compiler then waits for exactly the registers it needs; an
    // LDS-DMA fetch made every later LDS access wait for it)
    const bool xlive = g * 16 + li < p.B;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.skip + (long long)(p.clip0 + g * 16) * p.skip_bs), 0, (int)(((long long)(p.B - g * 16 < 16 ? p.B - g * 16 : 16)) * p.skip_bs * 4), 0x00020000);
    const int xoff = xlive ? (int)((long long)li * p.skip_bs * 4) + (wave * 64 + 8 * kq) * 4 : 0x7fff0000;
    f32x4 xr[2][2];                                           // [k-step][half]: floats 8 kq .. 8 kq + 3 / + 4 .. + 7 of the k-step
    auto fetch0 = [&](int t) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                xr[ks][hf] = bufload16(xrs, xlive ? xoff + (t * D + ks * 32 + hf * 4) * 4 : 0x7fff0000, 0);
    };
    // x[t] of clip li is scaled by its clip's 2^ex; the projection's rows (clips kq*4 + r) then go to the units of the recurrent
    // product (h travels as 2 h): * 2^-ex
    float xsc = 1.f, xcr[4] = {1.f, 1.f, 1.f, 1.f};
    if (fuse0) {
        const int cb = p.clip0 + g * 16, cl = p.clip0 + p.B - 1;
        xsc = s16_pow2(s16_exponent(*amax_at(pp.amax_x, cb + li <= cl ? cb + li : cl)));
#pragma unroll
        for (int r = 0; r < 4; ++r) xcr[r] = s16_pow2(LP16_HEXP - s16_exponent(*amax_at(pp.amax_x, cb + kq * 4 + r <= cl ? cb + kq * 4 + r : cl)));
    }
    // layer 0: accP = W_ih0 * x[t] from xr; trec >= 0: the requests for the peers' h[trec] and for x[t + 1] go out after the first
    // k-step
    // The operands of the projection that FOLLOWS the publish are made ready BEFORE it (x[t+1] converted to planes in the step's
    // head; layer 1: h0[t+1] validated there): a wait for an older load placed after the publish / output stores is a wait for
    // those stores' acknowledgements too (one in-order counter), 0.5-0.8 us in front of the projection's MFMAs.
    bf16x8 xa[2][2];
    auto convert0 = [&]() {
        bf16x8 (&a)[2][2] = xa;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
            const f32x4 s0 = xr[ks][0] * xsc, s1 = xr[ks][1] * xsc;
            const f16x4v h0 = __builtin_convertvector(s0, f16x4v), h1 = __builtin_convertvector(s1, f16x4v);
            const f16x4v l0 = __builtin_convertvector(s0 - __builtin_convertvector(h0, f32x4), f16x4v);
            const f16x4v l1 = __builtin_convertvector(s1 - __builtin_convertvector(h1, f32x4), f16x4v);
            a[ks][0] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
            a[ks][1] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto project0 = [&](int t, int trec, auto rec_tag) {
        bf16x8 (&a)[2][2] = xa;
#pragma unroll
        for (int n = 0; n < 4; ++n) accP[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // NO request below sits under a condition: a register that is loaded on one path and carried on the other becomes a
        // copy at the join -- and the copy waits for the load (the loop latch then waited for every request of the step)
        if constexpr (decltype(rec_tag)::value && !LATE) load_a(hmine, trec, arec);
        mac(a, wb, accP, 0, 1);
        mac(a, wb, accP, 1, 2);
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) accP[n][r] *= xcr[r];
        if constexpr (decltype(rec_tag)::value && LATE) {
            __builtin_amdgcn_sched_barrier(0);
            await_publish(trec);
            load_a(hmine, trec, arec);
        }
        (void)t;
    };
    auto project = [&](int t) -> bool {                        // layer 1, polling path
        bf16x8 a[2][2];
        if (!load_valid(h0b, t, a)) return false;
#pragma unroll
        for (int n = 0; n < 4; ++n) accP[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        mac(a, wa, accP);
        return true;
    };
#pragma unroll
    for (int n = 0; n < 4; ++n) accP[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (layer == 1 && !project(0)) return;
    if (fuse0) {
        fetch0(0);
        convert0();
        project0(0, 0, std::false_type{});
        fetch0(p.T > 1 ? 1 : 0);                               // (leaves x[1] on its way)
    }
    // layer 1's skip value: requested without a divergent branch around the load (threads without a clip read clip0's row) --
    // the join of such a branch made the compiler wait for every outstanding request in the middle of the projection
    const float* skip_row = p.skip + (live ? erow : (long long)p.clip0) * p.skip_bs + eu;
    if (layer == 1) skip_next = skip_row[0];
    // layer 1: the operand of the projection of step t+1 (h0[t+1], written by the neighbouring XCD) is requested at the very end
    // of step t-1 (its registers are free once that step's projection has consumed them); it validates itself like every operand
    // (round 4, measured and not kept: TWO sets, h0[t+2] requested in front of step t's barrier like layer 0's x, the time loop
    //  unrolled by two -- 2.50 -> 2.54 us per step on the same box)
    bf16x8 ap[2][2];
    if (layer == 1) load_a(h0b, p.T > 1 ? 1 : 0, ap);

    // developer trace (AC_LSTM_DBG & 32): 100 MHz stamps of slice 0 / thread 0 of every role at steps 100 .. 103 into the (unused)
    // flag words of the control block: [role = 2 g + layer][step][point]
    const bool trc = AC_DEV_MODE(p.dbg, 32) && idx == 0 && tid == 0;
    unsigned long long* trw = reinterpret_cast<unsigned long long*>(p.ctl + LP_CTL_FLAGS) + (g * 2 + layer) * 64;
#define LP16_TRC(pt_) do { if (trc && t >= 100 && t < 104) trw[(t - 100) * 8 + (pt_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    auto step = [&](int t) -> bool {
        if (trc && t >= 100 && t < 104) { trw[(t - 100) * 8 + 6] = 0; trw[(t - 100) * 8 + 7] = 0; }
        LP16_TRC(0);
        if (trc && (t == 100 || t == 103)) trw[32 + (t == 103)] = __builtin_amdgcn_s_memtime();    // shader clock over three steps
        float gpre[4] = {bq[0], bq[1], bq[2], bq[3]};
        const float skipv = skip_next;
        if (live && layer == 0 && !FUSE) {
            const float* gp = p.gin0 + (long long)t * p.gin_ts + erow * (4 * D);
#pragma unroll
            for (int q = 0; q < 4; ++q) gpre[q] = gp[q * D + eu];
        }
        f32x4 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = accP[n];
        if (t > 0) {
            // (arec: requested in the previous step's projection; incomplete, or never requested without a projection: poll)
            const bool first_ok = (layer == 1 || fuse0) && (valid(arec) || AC_DEV_MODE(p.dbg, 4));
            if (trc && !first_ok) trw[63] += 1;                    // developer trace: steps whose early request came back incomplete
            if (!first_ok && !load_valid(hmine, t - 1, arec)) return false;
            LP16_TRC(1);
            if (layer == 0) mac(arec, wa, acc);
            else mac(arec, wb, acc);
        }
        bool ap_ok = true;
        if (fuse0) {
            convert0();                                        // x[t+1] (requested a step ago) -> planes
            // x[t+2] is requested HERE, in front of the barrier: its 32 KB per CU (as much as the exchange itself) then pass the CU's
            // 64 B/clk load path under the partial sums, the barrier and the gate arithmetic, when nothing else wants it.  Issued
            // behind the recurrent request in the projection (rounds 2-3) they shared the path with the peers' h: the projection
            // took 0.9-1.0 us instead of the 0.5 us of its MFMAs and layer 0 -- not layer 1 -- bound the kernel at 16 live clips
            // per group (2.5-2.8 us per step against 1.9 with one clip; profiles/r4_lstm_notes.md)
            fetch0(t + 2 < p.T ? t + 2 : p.T - 1);
        }
        // (layer 1's skip value is waited for at the output store, behind the publish stores: s_waitcnt vmcnt(0), i.e. their acknowledgements
        //  too.  Forcing its use HERE, where it has arrived, removes that wait; measured on one box, three alternating passes: 2.55 us per
        //  step either way -- the stores are acknowledged by the XCD's L2 before the projection's first MFMAs are through)
        if (layer == 1) ap_ok = valid(ap) || AC_DEV_MODE(p.dbg, 4);      // h0[t+1] (requested at the end of the previous step) complete?
        LP16_TRC(2);
        f32x4 (&pt)[8][16][LP16_PART_PITCH] = part[t & 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) pt[wave][kq * 4 + r][li] = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        lds_barrier();
        LP16_TRC(3);
        float hn = 0.f;
        if (gate_thr) {
            float pre[4];
            f32x4 pv[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) pv[w] = pt[w][ec][ej];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float sum = ((pv[0][q] + pv[1][q]) + (pv[2][q] + pv[3][q])) + ((pv[4][q] + pv[5][q]) + (pv[6][q] + pv[7][q]));
                pre[q] = __fmaf_rn(sum, wiv[q], gpre[q]);
            }
            const float ig = sigmoid_rcp(pre[0]), fg = sigmoid_rcp(pre[1]), gg = tanh_rcp(pre[2]), og = sigmoid_rcp(pre[3]);
            cstate = fg * cstate + ig * gg;
            hn = og * tanh_rcp(cstate);
            LP16_TRC(6);
            // ---- publish h[t] (no flag, no wait).  A non-finite state must not look like "not yet written": publish a finite
            // stand-in and record the step; lstm_tail_kernel turns this clip's outputs from that step on into NaN (above)
            const bool nonfinite = !(fabsf(hn) < 2.0f);
            if (nonfinite && live) atomicMin(pp.poison + erow, t);
            const float hp = nonfinite ? 0.f : hn;
            // h travels as 2 h = hi + lo with |hi| < 2 (bit 14 clear).  1.0 itself occurs when the gates saturate and fp16(2.0) has
            // bit 14 set: hi is capped at the largest fp16 below 2 and lo takes the rest (2.0 = 1.9990234375 + 2^-10 exactly; values
            // that merely round up to 2.0 keep an error of 2^-21 of 2 h).  (Unscaled h would need no cap but makes lo a denormal
            // for |h| < 0.25 and the kernel 5 % slower.)
            const float h2 = LP16_HSCALE * hp;
            _Float16 hh = (_Float16)h2;
            if (LP16_HEXP && fabsf((float)hh) >= 2.0f) hh = (_Float16)copysignf(1.9990234375f, h2);
            const _Float16 hl = (_Float16)(h2 - (float)hh);
            char* dst = hmine + (long long)t * p.h_ts + goff + (long long)idx * SLICE_BYTES;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, SLICE_BYTES, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hh), rs, hpos, 0, LP_SC0);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hl), rs, 512 + hpos, 0, LP_SC0);
            LP16_TRC(7);
            if (lane == 0) pubflag[wave] = (unsigned)t + 1u;    // (LATE waves: await_publish)
            if (layer == 0) {   // the copy layer 1 reads from the neighbouring XCD: kept for the batch store below
                hist[t % LP16_BATCH][0][tid] = __builtin_bit_cast(unsigned short, hh);
                hist[t % LP16_BATCH][1][tid] = __builtin_bit_cast(unsigned short, hl);
            }
        }
        LP16_TRC(4);
        if (layer == 0 && gate_thr && ((t + 1) % LP16_BATCH == 0 || t + 1 == p.T)) {   // hand the batch over (own slots: no barrier)
            for (int tb = t - (t % LP16_BATCH); tb <= t; ++tb) {
                const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(h0b + (long long)tb * p.h_ts + goff + (long long)idx * SLICE_BYTES), 0, SLICE_BYTES, 0x00020000);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) __builtin_amdgcn_raw_buffer_store_b16(hist[tb % LP16_BATCH][pl][tid], rx, pl * 512 + hpos, 0, LP_SC1);
            }
        }
        if (layer == 1) {
            if (live) {
                const float yv = hn + skipv;
                const long long o = erow * p.y_bs + (long long)t * D + eu;
                if (p.yout) p.yout[o] = yv;
                if (p.yout_elu) p.yout_elu[o] = elu1(yv);
            }
            // projection of step t+1 (the last step repeats its own: no request sits under a condition, see project0)
            const int t1 = t + 1 < p.T ? t + 1 : t;
            if (!ap_ok && !load_valid(h0b, t1, ap)) return false;   // layer 0 is steps ahead: normally complete
#pragma unroll
            for (int n = 0; n < 4; ++n) accP[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!LATE) load_a(hmine, t, arec);
            mac(ap, wa, accP, 0, 1);
            // cross-XCD / HBM round trips go BEHIND the recurrent request and have a whole step to arrive
            if constexpr (!LATE) skip_next = skip_row[(long long)t1 * D];
            mac(ap, wa, accP, 1, 2);
            if constexpr (LATE) {
                __builtin_amdgcn_sched_barrier(0);
                await_publish(t);
                load_a(hmine, t, arec);
                skip_next = skip_row[(long long)t1 * D];
            }
            load_a(h0b, t + 2 < p.T ? t + 2 : t1, ap);
        } else if (fuse0) {
            project0(t + 1 < p.T ? t + 1 : t, t, std::true_type{});
        }
        LP16_TRC(5);
        return true;
    };
    for (int t = 0; t < p.T; ++t)
        if (!step(t)) return;
#undef LP16_TRC
    };
    if (wave < 4) {
        if (layer_rt == 0) body(std::integral_constant<int, 0>{}, std::false_type{});
        else body(std::integral_constant<int, 1>{}, std::false_type{});
    } else {
        if (layer_rt == 0) body(std::integral_constant<int, 0>{}, std::true_type{});
        else body(std::integral_constant<int, 1>{}, std::true_type{});
    }
}

}  // namespace ac
