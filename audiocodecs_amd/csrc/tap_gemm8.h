// tap_gemm8: the split16 tap-GEMM of tap_gemm6.h (same arithmetic, same accumulation order per output element, same epilogue
// code: a layer's results are bit-identical whichever of the two runs it) with a different LOAD PIPELINE, for the long-contraction
// layers where the main loop is the kernel.
//
// What bound tap_gemm6's main loop (profiles/r3_tapgemm_trace.md item 5, round-4 ISA reading in profiles/r4_tapgemm8.md): the
// memory counter retires loads in issue order.  Its weight fragments go L2 -> registers one k-step ahead, so every k-step waits
// for a weight set that was requested AFTER the stage's activation loads -- and with it for those activation loads: an HBM round
// trip (2000 - 3700 cycles under load) has one stage (two k-steps of 768 matrix cycles) to arrive, and the wave parks.
// Here
//   * the WEIGHT stage (k = 32: [2 k-steps][BN / 32 column tiles][2 planes] fragments of 1 KiB, already in MFMA operand order in
//     the packed image) goes L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, issued from inline asm so that the compiler does not
//     drain it at every barrier), one stage ahead, into a two-slot ring; a wave reads its B fragments from there like its A
//     fragments.  Nothing waits for weights inside a stage;
//   * the ACTIVATION chunk (fp32 rows -> split16 planes, as in tap_gemm6) is requested TWO STAGES before the slab write that needs
//     it (one tap per chunk: two register sets; several taps: one set, requested in a chunk's last tap, written in the next
//     chunk's first tap);
//   * every stage has ONE wait -- "everything but this stage's own activation requests has landed" -- and one barrier.  The
//     requests of a stage are issued weights first, activations second, so the counted wait (vmcnt(A_SLOTS)) never touches the
//     newest activation chunk: its latency budget is two stages (>= 6000 cycles), whatever the tap count.
//   * 8 waves, 256 rows per tile (2 x 4 waves of 128 x 64 / 128 x 32): one workgroup per CU, every weight fragment enters the CU
//     once per 256 rows, the slab's halo rows amortise over twice the rows.
// Restrictions (run_tap falls back to tap_gemm6 otherwise): split16 arithmetic, ONE segment, taps inside one slab
// ((J - 1) * dil <= 7), N % 128 == 0.
#pragma once
#include "tap_gemm6.h"

namespace ac {

template <int WGM, int WGN, int WMT, int WN>
struct Tap8Cfg {
    using C6 = Tap6Cfg<WGM, WGN, WMT, WN, 7>;
    static_assert(WGM * WGN == 8, "8 waves");
    static constexpr int BM = C6::BM, BN = C6::BN, NT = C6::NT, A_ROWS = C6::A_ROWS, A_SLOTS = C6::A_SLOTS, PLANE = C6::PLANE;
    static constexpr int NTILES = BN / 32;                      // 32-column tiles of the workgroup
    static constexpr int B_PIECES = 2 * NTILES * 2;             // 1 KiB fragments per weight stage: [k-step][column tile][plane]
    static constexpr int B_SLOT = B_PIECES * 512;               // fp16 elements per ring slot
    static constexpr int PPW = B_PIECES / 8;                    // fragments each wave fetches per stage
    static_assert(B_PIECES % 8 == 0, "fragments divide over the 8 waves");
    static constexpr size_t a_bytes = (size_t)2 * 2 * PLANE * 2;        // two slabs x two planes
    static constexpr size_t b_bytes = (size_t)2 * B_SLOT * 2;           // two ring slots
    static_assert(a_bytes % 16 == 0, "ring base alignment");
    static constexpr size_t main_bytes = a_bytes + b_bytes;
    static constexpr size_t lds_bytes = main_bytes > C6::epi_bytes ? main_bytes : C6::epi_bytes;
    static_assert(lds_bytes <= 160 * 1024, "one workgroup per CU: slabs + weight ring (or the staged epilogue's tile) within 160 KB of LDS");
};

// one 1 KiB LDS-DMA transfer: lane l's 16 bytes at gsrc land at LDS byte address lds_dst + 16 l (lds_dst wave-uniform).  Inline asm:
// hipcc does not count it (cdna_hip_programming.md section 5.7) -- the caller waits with t8_wait<N>() and a barrier before any read.
__device__ __forceinline__ void t8_glds16(const void* gsrc, unsigned lds_dst) {
#ifdef T8_ABL_NODMA    // (ablation build, wrong results: no weight stage is fetched)
    return;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void t8_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// RM: row mode (a linear layer over a merged row matrix: one split16 scale per ROW instead of per clip) -- a compile-time copy of
// TapGemmParams::amax_rows, so that the per-slot scale registers exist only where they are needed
// J1: the segment has ONE tap (two activation register sets); otherwise J >= 2 (one set) -- see the main loop
// SPREAD: the stage's requests are issued between its MFMA units instead of at its top (round 5, `tasks` below)
template <int WGM, int WGN, int WMT, int WN, bool RM, bool J1, bool SPREAD = true>
__global__ __launch_bounds__(512, 1) void tap_gemm8_kernel(const TapGemmParams p, const __bf16* __restrict__ wp) {
    using Cfg = Tap8Cfg<WGM, WGN, WMT, WN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, A_SLOTS = Cfg::A_SLOTS, PLANE = Cfg::PLANE, NTILES = Cfg::NTILES;
    constexpr int HALO = 7;
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (p.clk) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As0 = reinterpret_cast<__bf16*>(smem);                                       // [2 slabs][2 planes][A_ROWS][T6_PITCH]
    __bf16* Bs0 = reinterpret_cast<__bf16*>(reinterpret_cast<char*>(smem) + Cfg::a_bytes);   // [2 slots][B_PIECES][64 lanes][8]
    const unsigned bs_lds = (unsigned)(size_t)Bs0;                                       // LDS byte address of the ring

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int i32 = lane & 31, kh = lane >> 5;

    int id;
    {   // XCD-aware tile order (tap_gemm4.h)
        const int total = gridDim.x, q = total >> 3, r = total & 7, xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    }
    const int nt = id % p.ntiles; id /= p.ntiles;
    const int mt = id % p.mtiles;
    const int b = id / p.mtiles;
    const int m0 = mt * BM, n0 = nt * BN;
#ifdef T6_TRACE   // developer build: phase stamps of wave 1 (p.clk[4..15]) and stage stamps of every wave of ONE interior workgroup
    const bool t6_ph = p.clk && mt == p.mtiles / 2 && nt == 0 && b == p.B / 2 && tid == 64;
    const bool t8_on = p.clk && mt == p.mtiles / 2 && nt == 0 && b == p.B / 2 && lane == 0;
    int t8_stage = 0;
    if (t6_ph) p.clk[4] = clk_t0;
#define T8_STAMP(k) do { if (t8_on && t8_stage < 16) p.clk[16 + (wave * 16 + t8_stage) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    // unit stamps of stages 2 and 3 (one that writes a slab, one that requests a chunk when J = 2): word 5 / 6 of the stage slots 8 + u
#define T8_USTAMP(u) do { if (t8_on && (t8_stage == 2 || t8_stage == 3) && (u) < 8) p.clk[16 + (wave * 16 + 8 + (u)) * 8 + 3 + t8_stage] = __builtin_amdgcn_s_memtime(); } while (0)
    T6_PHASE(1);
#else
    const bool t6_ph = false;
#define T8_STAMP(k) do { } while (0)
#define T8_USTAMP(u) do { } while (0)
#endif

    // ---- the one segment (wave-uniform)
    const TapSeg& sg = p.seg[0];
    // (the clip's amax word is requested HERE and used ~1 000 cycles of index arithmetic later, where the scales are set: the ISA of round 4
    //  had the request and its s_waitcnt vmcnt(0) back to back in front of the tile's first weight request)
    const unsigned am_early = RM ? 0u : *amax_at(sg.amax, b);
    const int seg_J = sg.J, seg_Cw = sg.s * sg.cin, seg_kofs = sg.kofs, rowstep = sg.dil;
    const int NC = seg_Cw / KC;                                  // activation chunks; stages = NC * J
    const long long lo_t = (long long)m0 * sg.s - sg.pad;
    const long long hi_t = (long long)(m0 + BM - 1 + (sg.J - 1) * sg.dil) * sg.s + (sg.s - 1) - sg.pad;
    const bool inside = lo_t >= 0 && hi_t < sg.L;
    const bool seg_interior = inside || sg.s == 1;
    const int tsf = sg.s == 1 ? (int)sg.ts : sg.cin;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sg.x + (long long)b * sg.bs), 0,
                                                                             (int)(((long long)(sg.L - 1) * sg.ts + sg.cin) * 4), 0x00020000);
    const int R = BM + (sg.J - 1) * sg.dil;                      // slab rows in use
    // slot i of a thread = element tid + i * NT of the slab = (row tid / 8 + 64 i, 16-byte column tid % 8): the LDS offsets of a
    // thread's slots are 64 rows apart (one register + constants); only the last slot can fall outside the slab
    int a_boff[A_SLOTS];
    const int a_lds0 = (tid / (KC / 4)) * T6_PITCH + 4 * (tid % (KC / 4));
    const bool last_slot_ok = (tid + (A_SLOTS - 1) * NT) / (KC / 4) < Cfg::A_ROWS;
    unsigned a_zero = 0;
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
        const int e = tid + i * NT;
        const int row = e / (KC / 4), q = e % (KC / 4);
        long long t = (long long)(m0 + (row < R ? row : 0)) * sg.s - sg.pad;
        if (!inside) {
            if (sg.s == 1) {
                const long long jj = row < R ? src_index(sg, (int)t) : 0;
                if (jj < 0) a_zero |= 1u << i;
                t = jj < 0 ? 0 : jj;
            } else {
                t = 0;
            }
        }
        a_boff[i] = (int)((t * tsf + 4 * q) * 4);
    }
    // split16 scales: one per clip, or one per row of the merged row matrix (row mode)
    __builtin_amdgcn_sched_barrier(0);        // (the first use of am_early stays HERE, behind the index arithmetic above, not next to its request)
    float a_scale = 1.f, a_inv = 1.f;
    float a_rsc[RM ? A_SLOTS : 1];
    constexpr bool rowmode = RM;
    if (!RM) {
        // (through a VGPR-constrained asm: hipcc otherwise moves the wave-uniform word to an SGPR -- v_readfirstlane, and the wait with it --
        //  right behind the load)
        unsigned am = am_early;
        asm volatile("" : "+v"(am));
        const int se = s16_exponent(am);
        a_scale = s16_pow2(se);
        a_inv = s16_pow2(-se);
        a_rsc[0] = a_scale;
    } else {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int m = m0 + (tid + i * NT) / (KC / 4);
            a_rsc[RM ? i : 0] = s16_pow2(s16_exponent(sg.amax[m < p.M ? m : p.M - 1]));
        }
    }

    f32x16 acc[WMT][WN];
#pragma unroll
    for (int a = 0; a < WMT; ++a)
#pragma unroll
        for (int c = 0; c < WN; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    // A_SLOTS buffer loads of activation chunk `c` (k offset c * 32 inside the row) into ra; !live: through a descriptor of ZERO records
    // (every load returns zeros without touching memory) -- the COUNT of requests never depends on the path (see t8_wait below).
    // The per-slot offsets go to the loads as they are: a per-lane select (live ? offset : out-of-range) became v_cndmask writes into
    // registers the allocator had placed inside the loads' destination quads, and the write-after-write waits hipcc put in front of
    // them -- counted without the LDS-DMA requests it cannot see -- drained the weight ring at the top of every stage (round-4 ISA).
    constexpr int A_OOB = 0x7fff0000;
    const __amdgpu_buffer_rsrc_t a_rs_null = __builtin_amdgcn_make_buffer_rsrc((void*)(sg.x + (long long)b * sg.bs), 0, 0, 0x00020000);
    auto load_a = [&](int c, bool live, f32x4 (&ra)[A_SLOTS]) {
#ifdef T8_ABL_NOALOAD  // (ablation build, wrong results: every activation request goes through the zero-record descriptor)
        live = false;
#endif
        const int c_ = c * KC;
        if (seg_interior || !live) {
            const int soff = __builtin_amdgcn_readfirstlane(live ? c_ * 4 : 0);
            const __amdgpu_buffer_rsrc_t rs = live ? a_rs : a_rs_null;
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) ra[i] = bufload16(rs, a_boff[i], soff);
        } else {   // strided segment at a clip edge: every element through the padding rule
            int voff[A_SLOTS];
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) {
                const int e = tid + i * NT;
                const int row = e / (KC / 4), q = e % (KC / 4);
                const int cc = c_ + 4 * q;
                const int tp = sg.cin_shift >= 0 ? (cc >> sg.cin_shift) : (cc / sg.cin);
                const long long jj = row < R ? src_index(sg, (m0 + row) * sg.s + tp - sg.pad) : -1;
                voff[i] = jj < 0 ? A_OOB : (int)((jj * sg.ts + (cc - tp * sg.cin)) * 4);
            }
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) ra[i] = bufload16(a_rs, voff[i], 0);
        }
    };
    // the same, ONE slot (SPREAD: the requests of a stage sit between its MFMA units)
    auto load_slot = [&](int c, bool live, f32x4 (&ra)[A_SLOTS], int i) {
#ifdef T8_ABL_NOALOAD
        live = false;
#endif
        const int c_ = c * KC;
        if (seg_interior || !live) {
            const int soff = __builtin_amdgcn_readfirstlane(live ? c_ * 4 : 0);
            const __amdgpu_buffer_rsrc_t rs = live ? a_rs : a_rs_null;
            ra[i] = bufload16(rs, a_boff[i], soff);
        } else {
            const int e = tid + i * NT;
            const int row = e / (KC / 4), q = e % (KC / 4);
            const int cc = c_ + 4 * q;
            const int tp = sg.cin_shift >= 0 ? (cc >> sg.cin_shift) : (cc / sg.cin);
            const long long jj = row < R ? src_index(sg, (m0 + row) * sg.s + tp - sg.pad) : -1;
            ra[i] = bufload16(a_rs, jj < 0 ? A_OOB : (int)((jj * sg.ts + (cc - tp * sg.cin)) * 4), 0);
        }
    };
    // slot i of a landed chunk: split (split16.h) and written to the slab.  One slot at a time, so that the work can sit between the
    // MFMAs of a stage (compute's hook) instead of behind them
    auto store_slot = [&](const f32x4 (&ra)[A_SLOTS], __bf16* dst, int i) {
        // (the last slot's value is "used" by every thread: otherwise the compiler sinks that slot's LOAD into the branch below, only
        //  one wave then issues it, and the counted waits of the stage -- vmcnt(A_SLOTS) -- would let a weight fragment of the
        //  other waves stay in flight across the barrier)
        if (i + 1 == A_SLOTS) asm volatile("" ::"v"(ra[i]));
#ifdef T8_ABL_NOSTORE  // (ablation build, wrong results: no split, no slab write)
        return;
#endif
        if (i + 1 < A_SLOTS || last_slot_ok) {
            const f32x4 v = (a_zero & (1u << i)) ? f32x4{0.f, 0.f, 0.f, 0.f} : ra[i];
            split16_store4s(v, a_rsc[RM ? i : 0], dst, PLANE, a_lds0 + i * (NT / (KC / 4)) * T6_PITCH);
        }
    };
    auto store_a = [&](const f32x4 (&ra)[A_SLOTS], __bf16* dst) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) store_slot(ra, dst, i);
    };
    // weight stage s -> ring slot: this wave's PPW fragments.  Fragment f = (k-step ks, column tile t, plane pl) = (f / (2 NTILES), (f / 2) % NTILES, f % 2)
    const int ksteps = p.Ktot >> 4;
    const __bf16* wtile = wp + (long long)(n0 >> 5) * ksteps * 1024 + lane * 8;
    auto issue_b = [&](int c, int j, int slot) {
        const int k0 = (seg_kofs + j * seg_Cw + c * KC) >> 4;      // first k-step of the stage in the packed rows
#pragma unroll
        for (int i = 0; i < Cfg::PPW; ++i) {
            const int f = wave * Cfg::PPW + i;
            const int ks = f / (2 * NTILES), t = (f / 2) % NTILES, pl = f % 2;
            t8_glds16(wtile + ((long long)t * ksteps + k0 + ks) * 1024 + pl * 512, bs_lds + (unsigned)((slot * Cfg::B_PIECES + f) * 1024));
        }
    };

    auto issue_piece = [&](int c, int j, int slot, int i) {
        const int k0 = (seg_kofs + j * seg_Cw + c * KC) >> 4;
        const int f = wave * Cfg::PPW + i;
        const int ks = f / (2 * NTILES), t = (f / 2) % NTILES, pl = f % 2;
        t8_glds16(wtile + ((long long)t * ksteps + k0 + ks) * 1024 + pl * 512, bs_lds + (unsigned)((slot * Cfg::B_PIECES + f) * 1024));
    };

    // ---- fragment reads + MFMAs of one stage: slab `abuf`, tap row offset `jr`, ring slot `slot`
    const int a_frag = (wm * WMT * 32 + i32) * T6_PITCH + 8 * kh;
    // (s_setprio around the units' MFMAs / a static priority for waves 4..7: measured, inside the noise -- profiles/r4_tapgemm8.md)
    // Software pipeline over the 2 x WMT units (k-step, 32-row tile) of a stage: the A fragments of unit u + 1 (and, in a k-step's last
    // unit, the B fragments of the next k-step) are requested BEFORE unit u's MFMAs are issued, into the other of two fragment
    // buffers -- an LDS round trip (~150 cycles behind 8 waves' reads) travels under 3 WN MFMAs (96 WN cycles) instead of in front
    // of them.  sched_barrier keeps the blocks apart (hipcc would otherwise pull the reads back to their first use).
    auto compute = [&](int abuf, int jr, int slot, auto&& hook) {
        const __bf16* Ac = As0 + abuf * 2 * PLANE + a_frag + jr * T6_PITCH;
        const __bf16* Bc = Bs0 + slot * Cfg::B_SLOT + lane * 8;
        f16x8 bf[2][2][WN], af[2][2];                         // [buffer][plane]([column tile])
        auto read_b = [&](int ks, f16x8 (&dst)[2][WN]) {
#pragma unroll
            for (int c = 0; c < WN; ++c)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) dst[pl][c] = *reinterpret_cast<const f16x8*>(Bc + ((ks * NTILES + wn * WN + c) * 2 + pl) * 512);
        };
        auto read_a = [&](int ks, int a, f16x8 (&dst)[2]) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) dst[pl] = *reinterpret_cast<const f16x8*>(Ac + pl * PLANE + a * 32 * T6_PITCH + ks * 16);
        };
        read_b(0, bf[0]);
        read_a(0, 0, af[0]);
#pragma unroll
        for (int u = 0; u < 2 * WMT; ++u) {
            const int ks = u / WMT, a = u % WMT;
            if (u + 1 < 2 * WMT) read_a((u + 1) / WMT, (u + 1) % WMT, af[(u + 1) & 1]);
            if (a == WMT - 1 && ks == 0) read_b(1, bf[1]);
#pragma unroll
            for (int c = 0; c < WN; ++c) {   // lo hi, hi lo, hi hi: tap_gemm6's order
#ifdef T8_ABL_NOMFMA   // (ablation build, wrong results: the fragment reads stay, the MFMAs go)
                asm volatile("" ::"v"(af[u & 1][0]), "v"(af[u & 1][1]), "v"(bf[ks][0][c]), "v"(bf[ks][1][c]));
#else
                f32x16 v = acc[a][c];
                v = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u & 1][1], bf[ks][0][c], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u & 1][0], bf[ks][1][c], v, 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u & 1][0], bf[ks][0][c], v, 0, 0, 0);
#endif
            }
            hook(u);                                   // (same scheduling region as the unit's MFMAs: VALU / LDS-write work issues in their shadow)
            __builtin_amdgcn_sched_barrier(0);
            T8_USTAMP(u);
        }
    };
    auto no_hook = [](int) {};
    constexpr int UNITS = 2 * WMT;
    // the slab writes of a stage are spread over its LAST units, SPU slots per unit
    constexpr int SPU = (A_SLOTS + UNITS - 1) / UNITS, SU0 = UNITS - (A_SLOTS + SPU - 1) / SPU;
    auto store_hook = [&](const f32x4 (&ra)[A_SLOTS], __bf16* dst, bool st, int u) {
        if (u >= SU0 && st) {
#pragma unroll
            for (int k = 0; k < SPU; ++k)
                if ((u - SU0) * SPU + k < A_SLOTS) store_slot(ra, dst, (u - SU0) * SPU + k);
        }
    };

    // SPREAD (round 5).  With every request at the top of a stage, all eight waves spent its first 600 - 1 200 cycles pushing 9 vector-memory
    // instructions each through the CU's one address path -- no wave of a SIMD had an MFMA to issue -- and the younger wave of each pair
    // then ran the tail of the stage alone (s_memtime stamps, profiles/r5a_lockstep_trace.txt: a 256 x 256 stage of 3 072 matrix cycles per
    // SIMD took 4 650).  Now the stage's work items -- [weight pieces of stage s + 1][activation slots of chunk c + 2][slab writes of chunk
    // c + 1] -- are dealt over the stage's units IN THAT ORDER, one or two behind each unit's MFMAs: the first MFMA of a stage waits for
    // its fragment reads only.  The counted wait is unchanged: the weight pieces still precede the activation slots in program order.
    // Measured (profiles/r5c_spread_ab.txt, r5e_unit_trace.txt): 128 x 256 tiles -5.5 % (K = 4096), 256 x 256 within +-1 %, 256 x 128 with
    // seven taps +19 % -- run_tap picks per form.  The unit stamps show where a stage's time goes: a unit (6 MFMAs, 146 matrix ticks, 292
    // with the partner wave's) takes ~330 by itself, +150..400 on the wave that issues an LDS-DMA piece in it (the request arbitrates for
    // the LDS beside sixteen waves' fragment reads), +0..50 with a buffer load, ~+200 with a slab slot (15 VALU, 2 ds_write_b64).  Also
    // built and measured, not kept: the weight stage through registers (buffer loads + ds_write_b128: +3..7 % -- hipcc's own vmcnt waits
    // for the ring writes drain the activation requests, profiles/r5f_wreg_ab.txt) and the "ping-pong" loop (SIMD partners one segment
    // apart, load segments and pure-MFMA segments between workgroup barriers: +4..23 %, the load segments take 1 150 - 1 800 ticks against
    // a compute segment's 583 -- profiles/r5b_pingpong_ab.txt); the MFMA stream itself runs at 24.3 ticks per MFMA per SIMD at ANY
    // distance between dependent MFMAs and with LDS reads / VALU between them (tools/ubench/mfma_dep_distance.hip).
    auto tasks = [&](int u, bool has_next, int nc, int nj, int nslot, bool issue, int creq, f32x4 (&ra_req)[A_SLOTS], bool store, const f32x4 (&ra_st)[A_SLOTS], __bf16* dst,
                     auto issue_tag, auto store_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value, STORE = decltype(store_tag)::value;
        constexpr int NTASK = Cfg::PPW + (ISSUE ? A_SLOTS : 0) + (STORE ? A_SLOTS : 0);
        const int lo = u * NTASK / UNITS, hi = (u + 1) * NTASK / UNITS;
#pragma unroll
        for (int t = 0; t < NTASK; ++t) {
            if (t < lo || t >= hi) continue;
            if (t < Cfg::PPW) {
                if (has_next) issue_piece(nc, nj, nslot, t);
            } else if (ISSUE && t < Cfg::PPW + A_SLOTS) {
                load_slot(creq, issue, ra_req, t - Cfg::PPW);
            } else {
                if (store) store_slot(ra_st, dst, t - Cfg::PPW - (ISSUE ? A_SLOTS : 0));
            }
        }
    };

    using T = std::true_type;
    using F = std::false_type;
    // (the scales are in registers before the first LDS-DMA request goes out: the compiler's wait for ITS amax load would otherwise
    //  sit behind the weight requests it cannot see and drain them -- one exposed L2 round trip per tile)
#pragma unroll
    for (int i = 0; i < (RM ? A_SLOTS : 1); ++i) asm volatile("" ::"v"(a_rsc[i]));
#ifdef T6_TRACE
    T6_PHASE(2);
#endif
    if constexpr (J1) {
        // ---- ONE tap per chunk (linear layers, 1 x 1 convs): every stage is a new chunk.  Two register sets: chunk c + 2 is requested at
        // the top of stage c into the free set, chunk c + 1 (the other set, requested a stage earlier) is written to its slab at the
        // bottom.  The stage's single wait leaves exactly this stage's activation requests in flight: two stages of latency budget.
        f32x4 raA[A_SLOTS], raB[A_SLOTS];
        issue_b(0, 0, 0);
        load_a(0, true, raA);
        load_a(1, NC > 1, raB);
        store_a(raA, As0);
        t8_wait<A_SLOTS>();                 // (weights of stage 0 are older than chunk 0: landed; chunk 1 stays in flight)
        lds_barrier();
#ifdef T6_TRACE
        T6_PHASE(6);
#endif
        auto stage = [&](int c, f32x4 (&ra_next)[A_SLOTS], f32x4 (&ra_free)[A_SLOTS]) {
            T8_STAMP(0);
            const bool st = c + 1 < NC;
            __bf16* dst = As0 + ((c + 1) & 1) * 2 * PLANE;
            if constexpr (SPREAD) {
                T8_STAMP(1);
                compute(c & 1, 0, c & 1, [&](int u) { tasks(u, c + 1 < NC, c + 1, 0, (c + 1) & 1, c + 2 < NC, c + 2, ra_free, st, ra_next, dst, T{}, T{}); });
            } else {
                if (c + 1 < NC) issue_b(c + 1, 0, (c + 1) & 1);
                load_a(c + 2, c + 2 < NC, ra_free);
                __builtin_amdgcn_sched_barrier(0);
                T8_STAMP(1);
                // the slots of chunk c + 1 are split and written between the MFMAs of the stage's last A_SLOTS units
                compute(c & 1, 0, c & 1, [&](int u) { store_hook(ra_next, dst, st, u); });
            }
            __builtin_amdgcn_sched_barrier(0);
            T8_STAMP(2);
            t8_wait<A_SLOTS>();
            T8_STAMP(3);
            lds_barrier();
            T8_STAMP(4);
#ifdef T6_TRACE
            ++t8_stage;
#endif
        };
        for (int c = 0; c < NC; c += 2) {
            stage(c, raB, raA);
            if (c + 1 < NC) stage(c + 1, raA, raB);
        }
    } else {
        // ---- J >= 2 taps per chunk: ONE register set is enough for the same two-stage budget.  Chunk c + 1 is requested at the top of
        // the LAST tap of chunk c - 1 (behind that stage's weight requests, so the stage's wait -- vmcnt(A_SLOTS) -- leaves it in
        // flight) and written to its slab at the bottom of the FIRST tap of chunk c (that slab was last read in chunk c - 1); the
        // registers are free again for the request at the top of chunk c's last tap.
        f32x4 ra[A_SLOTS];
        issue_b(0, 0, 0);
        load_a(0, true, ra);
        store_a(ra, As0);
        load_a(1, NC > 1, ra);              // (chunk 1 has only the first stage to arrive: once per tile)
        t8_wait<A_SLOTS>();
        lds_barrier();
#ifdef T6_TRACE
        T6_PHASE(6);
#endif
        auto stage = [&](int c, int j, auto store_tag, auto issue_tag) {
            constexpr bool STORE = decltype(store_tag)::value, ISSUE = decltype(issue_tag)::value;
            const int s = c * seg_J + j;
            int nc = c, nj = j + 1;
            if (nj == seg_J) { nj = 0; nc = c + 1; }
            T8_STAMP(0);
            if constexpr (SPREAD) {
                T8_STAMP(1);
                const bool st = c + 1 < NC;
                __bf16* dst = As0 + ((c + 1) & 1) * 2 * PLANE;
                compute(c & 1, j * rowstep, s & 1, [&](int u) { tasks(u, nc < NC, nc, nj, (s + 1) & 1, c + 2 < NC, c + 2, ra, st, ra, dst, issue_tag, store_tag); });
            } else {
            if (nc < NC) issue_b(nc, nj, (s + 1) & 1);
            if (ISSUE) load_a(c + 2, c + 2 < NC, ra);
            __builtin_amdgcn_sched_barrier(0);
            T8_STAMP(1);
            if constexpr (STORE) {
                const bool st = c + 1 < NC;
                __bf16* dst = As0 + ((c + 1) & 1) * 2 * PLANE;
                compute(c & 1, j * rowstep, s & 1, [&](int u) { store_hook(ra, dst, st, u); });
            } else {
                compute(c & 1, j * rowstep, s & 1, no_hook);
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            T8_STAMP(2);
            if (ISSUE) t8_wait<A_SLOTS>(); else t8_wait<0>();
            T8_STAMP(3);
            lds_barrier();
            T8_STAMP(4);
#ifdef T6_TRACE
            ++t8_stage;
#endif
        };
        for (int c = 0; c < NC; ++c) {
            stage(c, 0, T{}, F{});
            for (int j = 1; j < seg_J - 1; ++j) stage(c, j, F{}, F{});
            stage(c, seg_J - 1, F{}, T{});
        }
    }

    tap6_epilogue<WGM, WGN, WMT, WN, HALO>(p, acc, smem, b, m0, n0, a_inv, rowmode, clk_t0, clk_r0, t6_ph);
#undef T8_STAMP
#undef T8_USTAMP
}

}  // namespace ac
