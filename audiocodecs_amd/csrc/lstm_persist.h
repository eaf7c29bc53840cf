// Persistent 2-layer LSTM for D = 512 on a full MI355X (256 CUs in 8 XCDs): ONE cooperative launch walks all T
// time steps; the per-step kernel of lstm.h (T+2 dependent launches, 12.6 us each) stays as the fallback for
// other widths / devices.  Same mathematics ([HF] EncodecLSTM :236-249, gate order i,f,g,o, + skip).
//
// Placement (measured with tools/ubench/xcd_exchange.hip): every CU hosts one workgroup; a workgroup reads its
// XCC_ID, takes a slot on its XCD and becomes
//     clip group g = xcd / 2          (16 clips: one MFMA M-tile; B <= 64 per launch)
//     layer      l = slot / 16        (each XCD hosts half of layer 0 and half of layer 1 of its group)
//     unit slice i = (xcd & 1) * 16 + slot % 16      -> hidden units 16 i .. 16 i + 15  (4 gates x 16 = 64 gate columns)
// so the 32 workgroups that exchange h_t of one (group, layer) sit 16 + 16 on an XCD pair: a publish-1KB /
// wait-32-flags / read-32KB round costs 2.3 us there (6 us when all 32 share one XCD, 12 us as a launch).
// The workgroup's weight slice (64 columns x K = 512: 128 KB per matrix) lives in REGISTERS for the whole
// sequence as MFMA B fragments (K split over the 4 waves, 4 gate tiles per wave); h_t is published in MFMA
// A-fragment order (a slice = one 1 KB k-step block), so consumers load straight into fragments; the cell state
// never leaves the thread that owns (clip, unit).
// Exchange protocol: payload stores and loads carry sc1 (agent scope: bypass the per-CU L1 / go through to
// where the other XCD sees them), `s_waitcnt vmcnt(0)` + workgroup barrier, then a relaxed agent-scope flag
// store; consumers poll the 32 flags with one wave.  No wbl2 / inv fences (those cost the 5-12 us of a grid
// barrier).  Every spin is bounded: on timeout a word is set and all workgroups leave.
// Layer 1 multiplies W_ih1 with h0[t+1] (layer 0 runs ahead) right after publishing h1[t], i.e. while its peers'
// slices travel: the input projection hides behind the exchange latency and needs no gin buffer.
#pragma once
#include "lstm_consts.h"
#include <hip/hip_runtime.h>
#include "lstm.h"

namespace ac {


struct LstmPersistParams {
    const float* gin0;     // [T][Ball][4D] layer-0 pre-activations (x W_ih0^T + b_ih0 + b_hh0), clip row stride 4D
    const float* w_pk;     // packed register images: [3 matrices: hh0, ih1, hh1][32 slices][4 waves][4 gates][8 ks][64 lanes][4]
    const float* bias1;    // [4D] b_ih1 + b_hh1
    float* hseq0;          // [T][groups][32 ks][64][4]  (hfrag_index order), time stride h_ts floats
    float* hseq1;
    const float* skip;     // module input x [Ball][T][D]: row (b, t) at skip + b*skip_bs + t*D
    float* yout;           // h1 + skip, raw (may be null)
    float* yout_elu;       // ELU(h1 + skip) (may be null)
    unsigned* ctl;         // control words, zeroed before the launch: slot counters, flags, timeout
    long long gin_ts, h_ts, skip_bs, y_bs;
    int B, T;              // clips of this launch (<= 64), steps
    int group0;            // first 16-clip group of this launch inside hseq (clip0 / 16)
    int clip0;             // first clip (for gin / skip / y rows)
    int dbg;               // developer timing modes (AC_LSTM_DBG; results invalid): 1 = layer 0 only, 2 = skip the MFMAs,
                           // 4 = do not wait for flags; 16 = test hook: every workgroup reports a timeout and leaves
};

// control block layout (32-bit words)
constexpr int LP_CTL_SLOTS = 0;                              // [8 XCDs] x 16 words
constexpr int LP_CTL_TIMEOUT = 8 * 16;                       // 1 word (+15 pad)
constexpr int LP_CTL_FLAGS = 9 * 16;                         // [4 groups][2 layers][32 slices] x LP_FLAG_STRIDE words
constexpr int LP_CTL_WORDS = LP_CTL_FLAGS + 4 * 2 * LP_SLICES * LP_FLAG_STRIDE;

typedef unsigned u32x4_lp __attribute__((ext_vector_type(4)));
#define LP_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int LP_SC1 = 16;   // buffer cache-policy bit: agent scope
constexpr int LP_SC0 = 1;    // buffer cache-policy bit: workgroup scope (a store is complete when the XCD's L2 has it)

__device__ __forceinline__ unsigned lp_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }   // hwreg(HW_REG_XCC_ID, 0, 4)

// Placement probe (run once at ac_finalize): does a 256-workgroup cooperative launch of this shape put exactly 32
// workgroups on each of 8 XCDs?  hist[x] counts the workgroups that read XCC_ID == x.
__global__ __launch_bounds__(256) void lstm_persist_probe_kernel(unsigned* hist) {
    if (threadIdx.x == 0) atomicAdd(&hist[lp_xcc_id() & 15], 1u);
}

constexpr int LP_NEVER = 0x7f7f7f7f;   // poison fill (hipMemsetAsync byte 0x7f): "this clip never went non-finite"

struct LstmTailParams {
    const unsigned* ctl;     // the launch's control words (slot counters, timeout word)
    unsigned* sticky;        // ST_* words
    const int* poison;       // [all clips] or null (exact-product kernel: NaN travels through the flags protocol unharmed)
    float* yout;             // [all clips][T][D] (may be null)
    float* yout_elu;
    long long y_bs;
    int clip0, B, T, D;      // the clips of this launch
    int xcds_used;           // 2 * ceil(B / 16): XCDs whose slot counter must read 32 (the others: <= 32)
};

// Runs right behind every persistent launch on the same stream (grid = (B, 4)).
//  * launch failed (bounded wait expired, or an XCD did not get its 32 workgroups): every output of the launch becomes
//    NaN -- never unwritten memory -- and a sticky word is raised that the next API call on the handle reports;
//  * a clip whose state went non-finite at step t0: its outputs from t0 on become NaN (what the reference computes).
__global__ __launch_bounds__(256) void lstm_tail_kernel(const LstmTailParams p) {
    const int b = blockIdx.x;
    const unsigned tmo = p.ctl[LP_CTL_TIMEOUT];
    bool placed = true;
    for (int x = 0; x < 8; ++x) {
        const unsigned n = p.ctl[LP_CTL_SLOTS + x * 16];
        placed = placed && (x < p.xcds_used ? n == 32u : n <= 32u);
    }
    const bool failed = tmo != 0u || !placed;
    if (failed && b == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_fetch_add(&p.sticky[(tmo == 2u || !placed) ? ST_LSTM_PLACEMENT : ST_LSTM_TIMEOUT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int t0 = p.T;
    if (failed) t0 = 0;
    else if (p.poison) {
        const int q = p.poison[p.clip0 + b];
        if (q < p.T) {
            t0 = q;
            if (blockIdx.y == 0 && threadIdx.x == 0)
                __hip_atomic_fetch_add(&p.sticky[ST_NONFINITE_CLIPS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (t0 >= p.T) return;
    const float nan = __uint_as_float(0x7fc00000u);
    const long long n = (long long)(p.T - t0) * p.D;
    const long long o = (long long)(p.clip0 + b) * p.y_bs + (long long)t0 * p.D;
    for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < n; i += 256LL * gridDim.y) {
        if (p.yout) p.yout[o + i] = nan;
        if (p.yout_elu) p.yout_elu[o + i] = nan;
    }
}

// all 32 flags of (group, layer) >= want ?  polled by wave 0; returns false on timeout
__device__ __forceinline__ bool lp_wait(unsigned* flags, unsigned want, unsigned* tmo, int lane, int dbg = 0) {
    if (AC_DEV_MODE(dbg, 4)) return true;
    for (unsigned spins = 0;; ++spins) {
        const unsigned f = lane < LP_SLICES ? __hip_atomic_load(&flags[lane * LP_FLAG_STRIDE], LP_RLX) : 0xffffffffu;
        if (__all(f >= want)) return true;
        if ((spins & 1023) == 1023 && __hip_atomic_load(tmo, LP_RLX)) return false;
        if (spins > (1u << 22)) { __hip_atomic_store(tmo, 1u, LP_RLX); return false; }
    }
}

__global__ __launch_bounds__(256) void lstm_persist_kernel(const LstmPersistParams p) {
    constexpr int D = LP_D;
    __shared__ float part[4][4][16][17];         // [wave = K quarter][gate][clip][unit]
    __shared__ unsigned s_x, s_slot;
    __shared__ int s_okp, s_okr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    unsigned* tmo = p.ctl + LP_CTL_TIMEOUT;
    if (tid == 0) {
        s_x = lp_xcc_id();
        s_slot = __hip_atomic_fetch_add(&p.ctl[LP_CTL_SLOTS + (s_x & 7) * 16], 1u, LP_RLX);
    }
    __syncthreads();
    const int x = s_x & 7, slot = s_slot;
    const int g = x >> 1;                                     // clip group of this launch
    const int G = (p.B + 15) >> 4;
    if (slot >= 32) {   // placement broken (see lstm_persist16.h)
        if (tid == 0) __hip_atomic_store(tmo, 2u, LP_RLX);
        return;
    }
    if (g >= G) return;
    if (AC_DEV_MODE(p.dbg, 1) && (slot >> 4) == 1) return;
    const int layer = slot >> 4, idx = (x & 1) * 16 + (slot & 15), u0 = idx * 16;

    // ---- weights -> registers: wave w holds k-steps 8w..8w+7 of the 4 gate tiles
    f32x4 wa[4][8], wb[4][8];                                 // layer 0: wa = W_hh0;  layer 1: wa = W_ih1, wb = W_hh1
    {
        const long long mat = (long long)LP_SLICES * 4 * 4 * 8 * 256;
        const float* base = p.w_pk + ((long long)idx * 4 + wave) * (4 * 8 * 256) + lane * 4;
        const float* pa = base + (layer == 0 ? 0 : mat);
        const float* pb = base + 2 * mat;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                wa[n][ks] = *reinterpret_cast<const f32x4*>(pa + (n * 8 + ks) * 256);
                wb[n][ks] = layer ? *reinterpret_cast<const f32x4*>(pb + (n * 8 + ks) * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    unsigned* flags0 = p.ctl + LP_CTL_FLAGS + ((g * 2 + 0) * LP_SLICES) * LP_FLAG_STRIDE;
    unsigned* flags1 = p.ctl + LP_CTL_FLAGS + ((g * 2 + 1) * LP_SLICES) * LP_FLAG_STRIDE;
    unsigned* myflag = (layer ? flags1 : flags0) + idx * LP_FLAG_STRIDE;
    float* hmine = layer ? p.hseq1 : p.hseq0;
    const long long goff = (long long)(p.group0 + g) * (D / 16) * 256;     // this group's 32 KB inside a time step

    // ---- the (clip, unit) cell this thread owns
    const int ec = tid >> 4, ej = tid & 15;
    const int eb = g * 16 + ec;                                // clip inside this launch
    const bool live = eb < p.B;
    const long long erow = (long long)(p.clip0 + eb);
    const int eu = u0 + ej;
    float cstate = 0.f;
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    if (layer == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = p.bias1[q * D + eu];
    }
    // where this thread's h lands inside the slice's 1 KB block (A-fragment order, see hfrag_index)
    const int hpos = (((ej >> 2) << 4) + ec) * 4 + (ej & 3);

    // A operand: this wave's 8 k-step blocks of h[t] of `seq` (sc1: through L2 / memory, never the per-CU L1)
    auto load_a = [&](const float* seq, int t, f32x4 (&a)[8]) {
        const float* src = seq + (long long)t * p.h_ts + goff + (long long)(wave * 8) * 256;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 8 * 1024, 0x00020000);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) a[ks] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (ks * 64 + lane) * 16, 0, LP_SC1));
    };
    // layer 1: accP = W_ih1 * h0[t]  (waits until layer 0 has published step t)
    f32x4 accP[4];
    auto project = [&](int t) -> bool {
        if (wave == 0) { const bool ok = lp_wait(flags0, (unsigned)(t + 1), tmo, lane, p.dbg); if (lane == 0) s_okp = ok; }
        __syncthreads();
        if (!s_okp) return false;
        f32x4 a[8];
        load_a(p.hseq0, t, a);
#pragma unroll
        for (int n = 0; n < 4; ++n) accP[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int n = 0; n < 4; ++n) accP[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][u], wa[n][ks][u], accP[n], 0, 0, 0);
        return true;
    };
#pragma unroll
    for (int n = 0; n < 4; ++n) accP[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (layer == 1 && !project(0)) return;

    for (int t = 0; t < p.T; ++t) {
        // epilogue operands early
        float gpre[4] = {bq[0], bq[1], bq[2], bq[3]};
        float skipv = 0.f;
        if (live) {
            if (layer == 0) {
                const float* gp = p.gin0 + (long long)t * p.gin_ts + erow * (4 * D);
#pragma unroll
                for (int q = 0; q < 4; ++q) gpre[q] = gp[q * D + eu];
            } else if (p.skip) {
                skipv = p.skip[erow * p.skip_bs + (long long)t * D + eu];
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = accP[n];           // layer 1: W_ih1 * h0[t], computed during the previous exchange
        // ---- recurrent term W_hh * h[t-1]: the critical path
        if (t > 0) {
            if (wave == 0) { const bool ok = lp_wait(layer ? flags1 : flags0, (unsigned)t, tmo, lane, p.dbg); if (lane == 0) s_okr = ok; }
            __syncthreads();
            if (!s_okr) return;
            f32x4 a[8];
            load_a(hmine, t - 1, a);
            if (AC_DEV_MODE(p.dbg, 2)) {
                acc[0] += a[0] + a[7];
            } else if (layer == 0) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][u], wa[n][ks][u], acc[n], 0, 0, 0);
            } else {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][u], wb[n][ks][u], acc[n], 0, 0, 0);
            }
        }
        // ---- K quarters meet in LDS (C layout: column = li = unit, rows kq*4 + r = clips)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][n][kq * 4 + r][li] = acc[n][r];
        __syncthreads();
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = gpre[q] + ((part[0][q][ec][ej] + part[1][q][ec][ej]) + (part[2][q][ec][ej] + part[3][q][ec][ej]));
        const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), og = sigmoidf_(pre[3]);
        cstate = fg * cstate + ig * gg;
        const float hn = og * tanhf_(cstate);
        // ---- publish h[t] (this slice's 1 KB block), then the flag
        {
            float* dst = hmine + (long long)t * p.h_ts + goff + (long long)idx * 256;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 1024, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn), rs, hpos * 4, 0, LP_SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the slice is at L2 / memory before the flag moves
        __syncthreads();                                       // also: `part` may be overwritten by the next step
        if (tid == 0) __hip_atomic_store(myflag, (unsigned)(t + 1), LP_RLX);
        if (layer == 1) {
            if (live) {                                        // module output, off the recurrent path
                const float yv = hn + skipv;
                const long long o = erow * p.y_bs + (long long)t * D + eu;
                if (p.yout) p.yout[o] = yv;
                if (p.yout_elu) p.yout_elu[o] = elu1(yv);
            }
            // next step's input projection while the peers' h1[t] slices travel
            if (t + 1 < p.T && !project(t + 1)) return;
        }
    }
}

}  // namespace ac
