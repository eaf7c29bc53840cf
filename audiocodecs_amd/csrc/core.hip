// Shared machinery of the library + the EnCodec encoder / decoder (see core.h for the translation-unit map).
#include "core.h"
#include "split16_kernels.h"
#include "lstm.h"
#include "lstm_persist.h"
#include "lstm_persist16.h"
#include "rvq.h"
#include "rvq16.h"
#include "tap_gemm4.h"
#include "tap_gemm6.h"
#include "tap_gemm8.h"
#include "thin.h"
#include "rb_fused.h"
#include "rb_fused6.h"
#include "thin_conv6.h"
#include "rb_fused6_128.h"
#include "enc_front.h"
#include "dec_tail.h"

namespace acimpl {


int fail(ac_handle* h, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}


// More than 64 KB of dynamic LDS must be opted into per kernel (and per device): once per handle.
int ensure_lds(ac_handle* h, const void* func, size_t bytes) {
    if (bytes <= 64 * 1024) return AC_OK;
    for (const void* f : h->lds_opted)
        if (f == func) return AC_OK;
    HIPCHK(h, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    h->lds_opted.push_back(func);
    return AC_OK;
}


Arch make_arch(const ac_config& c) {
    Arch a;
    const int F = c.num_filters, H = c.hidden_size, n = c.num_ratios;
    auto P = [](const char* part, int i, const char* rest) {
        return std::string(part) + ".layers." + std::to_string(i) + rest;
    };
    a.enc_stem = {P("encoder", 0, ".conv"), 0, 1, F, c.kernel_size, 1};
    int i = 1, ch = F;
    for (int r = n - 1; r >= 0; --r) {
        const int ratio = c.upsampling_ratios[r];
        const int hid = ch / c.compress;
        a.enc_rb3.push_back({P("encoder", i, ".block.1.conv"), 0, ch, hid, c.residual_kernel_size, 1});
        a.enc_rb1.push_back({P("encoder", i, ".block.3.conv"), 0, hid, ch, 1, 1});
        a.enc_rbs.push_back({P("encoder", i, ".shortcut.conv"), 0, ch, ch, 1, 1});
        a.enc_down.push_back({P("encoder", i + 2, ".conv"), 0, ch, 2 * ch, 2 * ratio, ratio});
        i += 3;
        ch *= 2;
    }
    a.D = ch;
    a.enc_lstm = P("encoder", i, ".lstm");
    a.enc_final = {P("encoder", i + 2, ".conv"), 0, ch, H, c.last_kernel_size, 1};
    a.dec_first = {P("decoder", 0, ".conv"), 0, H, ch, c.kernel_size, 1};
    a.dec_lstm = P("decoder", 1, ".lstm");
    i = 2;
    for (int r = 0; r < n; ++r) {
        const int ratio = c.upsampling_ratios[r];
        a.dec_up.push_back({P("decoder", i + 1, ".conv"), 1, ch, ch / 2, 2 * ratio, ratio});
        const int c2 = ch / 2, hid = c2 / c.compress;
        a.dec_rb3.push_back({P("decoder", i + 2, ".block.1.conv"), 0, c2, hid, c.residual_kernel_size, 1});
        a.dec_rb1.push_back({P("decoder", i + 2, ".block.3.conv"), 0, hid, c2, 1, 1});
        a.dec_rbs.push_back({P("decoder", i + 2, ".shortcut.conv"), 0, c2, c2, 1, 1});
        i += 3;
        ch = c2;
    }
    a.dec_head = {P("decoder", i + 1, ".conv"), 0, ch, 1, c.last_kernel_size, 1};
    return a;
}


int prof_name(ac_handle* h, const char* nm) {
    for (size_t i = 0; i < h->prof_names.size(); ++i)
        if (h->prof_names[i] == nm) return (int)i;
    h->prof_names.push_back(nm);
    return (int)h->prof_names.size() - 1;
}


hipEvent_t next_event(ac_handle* h) {
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_used++];
}

// bind a pool (device memory of pool_bytes(pool_B, rows) bytes, 256-byte aligned; null: no pool -- entry points that launch
// nothing in split-operand arithmetic)
void pool_bind(ac_handle* h, void* mem, int pool_B, size_t rows) {
    h->amax_buf = reinterpret_cast<unsigned*>(mem);
    h->amax_B = mem ? std::max(pool_B, 1) : 0;
    h->amax_next = 0;
    h->row_buf = mem ? h->amax_buf + (size_t)AMAX_SLOTS * h->amax_B * AMAX_STRIDE : nullptr;
    h->row_cap = mem ? 8 * align_up(std::max<size_t>(rows, 64), 64) : 0;
    h->row_next = 0;
}


// `words` 32-bit words from p (16-byte aligned, words % 4 == 0) to zero with zero16_kernel (split16_kernels.h: not a memset node)
static void zero_words(hipStream_t st, unsigned* p, size_t words) {
    const size_t n16 = words / 4;
    const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((n16 + 255) / 256, 1), 2048);
    hipLaunchKernelGGL(zero16_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<s16_f32x4*>(p), n16);
}


// start of a pass over B clips: all slots of the bound pool back to zero (one memset of AMAX_SLOTS x B lines on the caller's stream)
int amax_begin(ac_handle* h, hipStream_t st, int B) {
    h->amax_next = 0;
    if (h->gemm_fp32 || !h->amax_buf) return AC_OK;
    if (B > h->amax_B) return fail(h, AC_ENOMEM, "workspace pool holds amax slots for %d clips, the pass has %d", h->amax_B, B);
    zero_words(st, h->amax_buf, (size_t)AMAX_SLOTS * h->amax_B * AMAX_STRIDE);
    HIPCHK(h, hipGetLastError());
    return AC_OK;
}

// a fresh slot for a producer's output (null when the arithmetic does not use them)
unsigned* amax_new(ac_handle* h) {
    if (h->gemm_fp32 || !h->amax_buf || h->amax_next >= AMAX_SLOTS) return nullptr;
    return h->amax_buf + (size_t)(h->amax_next++) * h->amax_B * AMAX_STRIDE;
}

// the amax of a tensor a consumer is about to split: the producer's, or one more read of the tensor
const unsigned* amax_of(ac_handle* h, hipStream_t st, const float* x, long long bs, long long ts, int L, int C, int B, const unsigned* known) {
    if (known) return known;
    if (B > h->amax_B) return nullptr;
    unsigned* slot = amax_new(h);
    if (!slot) return nullptr;
    const long long n = (long long)L * C;
    const int gx = (int)std::max<long long>(1, std::min<long long>((n / 4 + 256 * 8 - 1) / (256 * 8), std::max(4, 8192 / std::max(1, B))));   // >= 8 vectors per thread, <= 8192 workgroups
    ProfScope ps(h, st, "amax_kernel", 0.0, (double)B * n * 4.0);
    hipLaunchKernelGGL(amax_kernel, dim3(gx, B), dim3(256), 0, st, x, bs, ts, L, C, slot);
    return slot;
}


// `rows` words of the row ring (split16.h row mode); zeroed when a kernel is going to atomicMax into them.  Null when the
// bound pool's ring is too small for `rows` (the workspace planners size it for the widest row matrix of the pass)
unsigned* rowmax_new(ac_handle* h, hipStream_t st, long long rows, bool zero) {
    const size_t need = ((size_t)rows + 63) / 64 * 64;      // allocations are 64-word granules; the ring holds eight of the largest
    if (!h->row_buf || need > h->row_cap) return nullptr;
    if (h->row_next + need > h->row_cap) h->row_next = 0;
    unsigned* r = h->row_buf + h->row_next;
    h->row_next += need;
    if (zero) {          // (the granule is 64 words: whole 16-byte units, the tail of the last granule belongs to this allocation too)
        zero_words(st, r, need);
        if (hipGetLastError() != hipSuccess) return nullptr;
    }
    return r;
}


// slot holding the bound amax(x) + add of a tensor y with |y| <= |x| + add (LSTM with skip: |h| < 1); null when x has no amax
const unsigned* amax_plus(ac_handle* h, hipStream_t st, const Act& x, float add, int B) {
    if (!x.amax || x.amax_n != B) return nullptr;
    unsigned* slot = amax_new(h);
    if (!slot) return nullptr;
    hipLaunchKernelGGL(amax_add_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, x.amax, add, slot, B);
    return slot;
}


template <int WGM, int WGN, int WM, int WN, bool VEC>
void launch_tap(const TapGemmParams& p0, hipStream_t st) {
    TapGemmParams p = p0;
    constexpr int BM = WGM * WM * 16, BN = WGN * WN * 16;
    p.mtiles = cdiv(p.M, BM);
    p.ntiles = cdiv(p.N, BN);
    const size_t lds = tap_gemm_lds_bytes<WGM, WGN, WM, WN>();
    const long long blocks = (long long)p.B * p.mtiles * p.ntiles;
    hipLaunchKernelGGL((tap_gemm_kernel<WGM, WGN, WM, WN, VEC>), dim3((unsigned)blocks), dim3(WGM * WGN * 64), lds, st, p);
}


template <int WGM, int WGN, int WM, int WN>
int launch_tap4(ac_handle* h, const TapGemmParams& p0, hipStream_t st) {
    using Cfg = Tap4Cfg<WGM, WGN, WM, WN>;
    TapGemmParams p = p0;
    p.mtiles = cdiv(p.M, Cfg::BM);
    p.ntiles = cdiv(p.N, Cfg::BN);
    if (int rc = ensure_lds(h, reinterpret_cast<const void*>(tap_gemm4_kernel<WGM, WGN, WM, WN>), Cfg::lds_bytes)) return rc;
    const long long blocks = (long long)p.B * p.mtiles * p.ntiles;
    const size_t lds = Cfg::lds_bytes;
    hipLaunchKernelGGL((tap_gemm4_kernel<WGM, WGN, WM, WN>), dim3((unsigned)blocks), dim3(Cfg::NT), lds, st, p);
    return AC_OK;
}


// One segment of the A operand for a conv reading `x` (time steps of C channels).  Inputs are
// already activated by their producer (TapGemmParams::y_elu), so no segment carries ELU.
// `left` < 0: causal (all (J-1)*s padding steps on the left); otherwise `left` / `right` padding steps (non-causal SEANet of
// WavTokenizer: right = total/2, left = total - right; `extra` completes the last frame on the right in both cases).
TapSeg make_seg(const Act& x, int s, int J, int pad /*PAD_**/, int extra, int kofs, const float* rel_len, int left, int right) {
    TapSeg g{};
    g.x = x.p;
    g.amax = x.amax;
    g.amax_n = x.amax_n;
    g.bs = x.bs;
    g.ts = x.ts;
    g.rel_len = rel_len;
    g.L = x.L;
    g.cin = x.C;
    g.cin_shift = -1;
    for (int sh = 0; sh < 30; ++sh)
        if ((1 << sh) == x.C) g.cin_shift = sh;
    g.s = s;
    g.J = J;
    const int pad_left = left < 0 ? (J - 1) * s : left;
    const int max_pad = std::max(pad_left, right + extra);
    g.Lp = (pad == PAD_REFLECT && x.L <= max_pad) ? max_pad + 1 : x.L;
    g.lim = pad != PAD_ZERO ? x.L + right + extra : x.L;
    g.reflect = pad;
    g.elu = 0;
    g.kofs = kofs;
    g.pad = pad_left;
    g.dil = 1;
    return g;
}


int run_tap(ac_handle* h, hipStream_t st, TapGemmParams& p) {
    bool vec = (p.Ktot % 4 == 0) && aligned16(p.w);
    bool fast = vec && (p.N % 4 == 0) && (p.y_rs % 4 == 0) && (p.y_bs % 4 == 0) && (!p.y || aligned16(p.y)) &&
                (!p.y_elu || aligned16(p.y_elu)) && (long long)p.N * p.Ktot * 4 < (1LL << 31);
    for (int i = 0; i < p.nseg; ++i) {
        const TapSeg& s = p.seg[i];
        vec = vec && (s.cin % 4 == 0) && (s.ts % 4 == 0) && (s.bs % 4 == 0) && (s.kofs % 4 == 0) && aligned16(s.x);
        fast = fast && ((s.s * s.cin) % KC == 0) && (s.ts == s.cin || s.s == 1) && !s.rel_len && !s.elu &&
               ((long long)(s.L - 1) * s.ts + s.cin) * 4 < (1LL << 31);
        if (s.J > 8) return fail(h, AC_EINVAL, "conv with %d taps exceeds the kernel limit of 8", s.J);
    }
    fast = fast && vec;
    double kk = 0, inb = 0;
    for (int i = 0; i < p.nseg; ++i) {
        kk += (double)p.seg[i].J * p.seg[i].s * p.seg[i].cin;
        inb += (double)p.B * p.seg[i].L * p.seg[i].cin * 4.0;
    }
    const double flops = 2.0 * p.B * (double)p.M * p.N * kk;
    const double bytes = inb + (double)p.B * p.M * p.N * 4.0 * ((p.y ? 1 : 0) + (p.y_elu ? 1 : 0)) + (double)p.N * p.Ktot * 4.0;
    // split-operand kernel on the bf16 pipe (tap_gemm6.h) where the shape allows and the weights were packed for it
    const __bf16* w6 = nullptr;
    if (fast && !h->gemm_fp32 && (p.N % 64 == 0 || p.N % 96 == 0)) {
        auto it = h->w6_of.find((size_t)(p.w - h->blob));
        bool ok6 = it != h->w6_of.end();
        for (int i = 0; ok6 && i < p.nseg; ++i) ok6 = p.seg[i].kofs % 32 == 0;
        if (ok6) w6 = reinterpret_cast<const __bf16*>(h->blob + it->second);
    }
    int rc = AC_OK;
    char shape[64] = "";
    if (h->prof && h->prof_detail)
        std::snprintf(shape, sizeof shape, " B%d M%d N%d K%d J%d s%d%s", p.B, p.M, p.N, (int)kk, p.seg[0].J, p.seg[0].s, p.seg[0].dil == 1 ? "" : p.seg[0].dil == 3 ? " d3" : " d9");
#define TAP_CASE(WGM, WGN, WM, WN)                                                                          \
    do {                                                                                                    \
        if (fast) {                                                                                         \
            ProfScope ps(h, st, (std::string("tap_gemm4_kernel<" #WGM ", " #WGN ", " #WM ", " #WN ">") + shape).c_str(), flops, bytes); \
            rc = launch_tap4<WGM, WGN, WM, WN>(h, p, st);                                                   \
        } else if (vec) {                                                                                   \
            ProfScope ps(h, st, "tap_gemm_kernel<" #WGM ", " #WGN ", " #WM ", " #WN ", true>", flops, bytes); \
            launch_tap<WGM, WGN, WM, WN, true>(p, st);                                                      \
        } else {                                                                                            \
            ProfScope ps(h, st, "tap_gemm_kernel<" #WGM ", " #WGN ", " #WM ", " #WN ", false>", flops, bytes); \
            launch_tap<WGM, WGN, WM, WN, false>(p, st);                                                     \
        }                                                                                                   \
    } while (0)
    const bool want_rows = p.amax_out_rows != nullptr;    // (any non-null value is a request)
    p.amax_out_rows = nullptr;
    const int want_rowmode = p.amax_rows;
    p.amax_rows = 0;
    if (w6) {
        p.clk = h->clk_dev;
#ifdef T6_TRACE   // developer trace builds only: clock / stamps of ONE layer shape "M,N,K"
        if (const char* ts = std::getenv("AC_TRACE_SHAPE")) {
            int tm = 0, tn = 0, tk = 0;
            if (std::sscanf(ts, "%d,%d,%d", &tm, &tn, &tk) == 3 && !((tm == 0 || tm == p.M) && tn == p.N && tk == (int)kk)) p.clk = nullptr;   // (M = 0: any M)
        }
#endif
        auto iv = h->winv_of.find((size_t)(p.w - h->blob));
        // (row mode only on the caller's request -- the linear layers over merged token matrices: a conv that merely happens to
        // run with one clip must scale like the same conv in a batch, or a clip's result would depend on the batch size)
        const bool rowmode = want_rowmode && iv != h->winv_of.end() && p.B == 1 && p.nseg == 1 && p.seg[0].J == 1 && p.seg[0].s == 1 && p.seg[0].pad == 0 &&
                             p.seg[0].lim >= p.M && p.seg[0].L >= p.M && p.y_off == 0;
        if (rowmode) {                    // split16.h row mode: a linear layer over a merged row matrix -- one scale per row
            TapSeg& sg = p.seg[0];
            if (!(sg.amax && sg.amax_n == -p.M)) {
                unsigned* rm = rowmax_new(h, st, p.M, false);
                if (!rm) return fail(h, AC_ENOMEM, "the workspace pool's row ring is too small for %d rows", p.M);
                ProfScope ps(h, st, "rowmax_kernel", 0.0, (double)p.M * sg.cin * 4.0);
                hipLaunchKernelGGL(rowmax_kernel, dim3((unsigned)cdiv(p.M, 4)), dim3(256), 0, st, sg.x, sg.ts, (long long)p.M, sg.cin, rm);
                sg.amax = rm;
            }
            p.amax_rows = 1;
            p.winv = h->blob + iv->second;
            if (want_rows) {              // the caller asked for the output's row words: a fresh, zeroed array
                p.amax_out_rows = rowmax_new(h, st, p.M, true);
                if (!p.amax_out_rows) return fail(h, AC_ENOMEM, "the workspace pool's row ring is too small for %d rows", p.M);
            }
        } else if (iv != h->winv_of.end()) {     // split16.h: every operand tensor needs its amax; the output reports its own
            for (int i = 0; i < p.nseg; ++i) {
                TapSeg& sg = p.seg[i];
                sg.amax = amax_of(h, st, sg.x, sg.bs, sg.ts, sg.L, sg.cin, p.B, sg.amax_n == p.B ? sg.amax : nullptr);
                if (!sg.amax) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
            }
            p.winv = h->blob + iv->second;
            p.amax_out = amax_new(h);
            // plain conv outputs store straight from the accumulators (tap_gemm6.h); AC_TAP_EPI=staged: the LDS-staged epilogue
            const bool staged_env = h->dev.tap_epi_staged != 0;       // (ac_debug_set "tap_epi_staged": a test flips it)
            // (ELU flavour without a residual, Snake flavour with or without one: the combinations the four codecs produce)
            p.epi_direct = !staged_env && !p.gelu && !p.scale && !p.tanh_out && p.y_off == 0 && p.y_len == 0 &&
                           (!p.res || (p.alpha && (long long)p.M * p.res_rs * 4 < 0x7fffffffLL && p.res_rs * 4 < (1 << 20))) && (!p.alpha || p.y_elu) &&
                           (p.n_valid == 0 || p.n_valid == p.N) && (p.alpha ? p.N < 128 && p.N % 32 == 0 : p.N % 128 == 0) &&      // (measured: Snake / residual layers of 128+ channels are faster through the LDS-staged 16-byte rows)
                           (long long)p.M * p.y_rs * 4 < 0x7fffffffLL && p.y_rs * 4 < (1 << 20);
        }
        p.stagger = h->dev.tap_stagger;
        const bool dil_env = h->dev.tap_dil != 0;                 // developer / tests: 0 -> slab reload per tap
        const bool dil_slab = dil_env && !p.amax_rows && p.nseg == 1 && p.seg[0].dil != 1   /* (the wide-slab instantiation has no row mode: CAN_ROWMODE, tap_gemm6.h) */ && p.seg[0].s == 1 && (p.seg[0].J - 1) * p.seg[0].dil <= T6_DIL_HALO;
#define TAP6_LAUNCH(WGM, WGN, WMT, WN, NP)                                                                              \
    do {                                                                                                                \
        if ((rc = ensure_lds(h, reinterpret_cast<const void*>(tap_gemm6_kernel<WGM, WGN, WMT, WN, NP>), Cfg6::lds_for(NP)))) return rc; \
        ProfScope ps(h, st, (std::string("tap_gemm6_kernel<" #WGM ", " #WGN ", " #WMT ", " #WN ", " #NP ">") + shape).c_str(), flops, bytes); \
        hipLaunchKernelGGL((tap_gemm6_kernel<WGM, WGN, WMT, WN, NP>), dim3((unsigned)blocks), dim3(Cfg6::NT), Cfg6::lds_for(NP), st, p, w6); \
    } while (0)
#define TAP6_CASE(WGM, WGN, WMT, WN)                                                                                    \
    do {                                                                                                                \
        using Cfg6 = Tap6Cfg<WGM, WGN, WMT, WN>;                                                                        \
        p.mtiles = cdiv(p.M, Cfg6::BM);                                                                                 \
        p.ntiles = p.N / Cfg6::BN;                                                                                      \
        const long long blocks = (long long)p.B * p.mtiles * p.ntiles;                                                  \
        if (p.winv && dil_slab && (WN >= 2 || (WGM == 1 && WGN == 4 && kk >= 2048))) {                                  \
            /* dilated taps out of one wide slab (tap_gemm6.h: T6_DIL_HALO); measured per arrangement on DAC's layers: the 128 x 32  \
               tile loses its third workgroup per CU to the larger slab and gains only for long contractions, 64 x 32 tiles lose */ \
            using Cfg6D = Tap6Cfg<WGM, WGN, WMT, WN, T6_DIL_HALO>;                                                      \
            if ((rc = ensure_lds(h, reinterpret_cast<const void*>(tap_gemm6_kernel<WGM, WGN, WMT, WN, 2, T6_DIL_HALO>), Cfg6D::lds_for(2)))) return rc; \
            ProfScope ps(h, st, (std::string("tap_gemm6_kernel<" #WGM ", " #WGN ", " #WMT ", " #WN ", 2, dil>") + shape).c_str(), flops, bytes); \
            hipLaunchKernelGGL((tap_gemm6_kernel<WGM, WGN, WMT, WN, 2, T6_DIL_HALO>), dim3((unsigned)blocks), dim3(Cfg6D::NT), Cfg6D::lds_for(2), st, p, w6); \
        } else TAP6_LAUNCH(WGM, WGN, WMT, WN, 2);                                                                       \
    } while (0)
        // tap_gemm8.h: the 256-row, 8-wave kernel with the weight stage through an LDS-DMA ring and activation chunks requested two
        // chunks ahead -- one segment, taps inside one slab, N % 128 == 0 (the same arithmetic in the same order: bit-identical outputs)
        {
            const TapSeg& s0 = p.seg[0];
            const bool can8 = p.winv && p.nseg == 1 && (s0.J - 1) * s0.dil <= 7 && !(s0.dil != 1 && s0.s != 1) && p.N % 128 == 0 && (s0.s * s0.cin) % 32 == 0 &&
                              (!p.epi_direct || p.N % 128 == 0);
            const int want8 = h->dev.tap8;       // 0: never, 1: wherever the shape allows (developer A/B), -1: cost model
            bool use8 = false;
            if (can8 && want8 != 0) {
                // tile forms (8 waves each): 1 = 256 x 256 (2 x 4 waves of 128 x 64), 3 = 128 x 256 (2 x 4 waves of 64 x 64) where 256-row
                // tiles would leave CUs idle or rows empty (M = 750: three tiles per clip; M = 125), 2 = 256 x 128 (4 x 2 waves of 64 x 64) for
                // N % 256 != 0.  Score = rate relative to form 1 (EnCodec / Mimi / DAC layers, profiles/r4_tapgemm8.md) x how evenly the
                // workgroups fill the 256 CUs x the share of tile rows that exist.
                auto fill8 = [&](int bm, int bn) {
                    const double w = (double)p.B * cdiv(p.M, bm) * (p.N / bn) / 256.0;
                    return w / std::ceil(w) * ((double)p.M / ((double)cdiv(p.M, bm) * bm));
                };
                const double sc1 = p.N % 256 == 0 ? 1.00 * fill8(256, 256) : 0.0;
                const double sc3 = p.N % 256 == 0 ? 0.90 * fill8(128, 256) : 0.0;
                int form = sc1 >= sc3 ? 1 : 3;
                bool model = (form == 1 ? sc1 : sc3) >= 0.80;
                if (p.N % 256 != 0) {      // 64 x 64 wave tiles over 128 columns lose to tap_gemm6's three workgroups per CU except on long contractions
                    form = 2;
                    model = kk >= 3072 && fill8(256, 128) >= 0.70;
                }
                // tiny launches (the batch-1 / batch-8 regime: at most 64 tiles of 128 x 128): every workgroup has a CU of its own and walks its K
                // loop at one memory round trip per stage -- the ring's deeper look-ahead is what counts (1.39 -> 1.34 ms per 1 s call)
                if ((double)p.B * cdiv(p.M, 128) * (p.N / 128) <= 64.0) {
                    form = p.N % 256 == 0 ? 3 : 2;
                    model = true;
                }
                if (h->dev.tap8_form >= 1 && h->dev.tap8_form <= 3 && (h->dev.tap8_form == 2 || p.N % 256 == 0)) form = h->dev.tap8_form;
                use8 = want8 >= 1 || model;
                if (use8) {
#define TAP8_LAUNCH_KP(WGM, WGN, WMT, WN, RM, J1, SP)                                                                    \
    do {                                                                                                                \
        if ((rc = ensure_lds(h, reinterpret_cast<const void*>(tap_gemm8_kernel<WGM, WGN, WMT, WN, RM, J1, SP>), Cfg8::lds_bytes))) return rc; \
        hipLaunchKernelGGL((tap_gemm8_kernel<WGM, WGN, WMT, WN, RM, J1, SP>), dim3((unsigned)blocks), dim3(Cfg8::NT), Cfg8::lds_bytes, st, p, w6); \
    } while (0)
#define TAP8_LAUNCH_K(WGM, WGN, WMT, WN, RM, J1)                                                                         \
    do {                                                                                                                \
        /* requests dealt between the MFMA units (tap_gemm8.h SPREAD): measured per form -- 128 x 256 tiles gain, 256 x 256 are level, 256 x 128 lose */ \
        const bool spread = h->dev.tap8_spread == 2 || (h->dev.tap8_spread == 1 && (WGM) * (WMT) * 32 == 128);         \
        if (spread) TAP8_LAUNCH_KP(WGM, WGN, WMT, WN, RM, J1, true); else TAP8_LAUNCH_KP(WGM, WGN, WMT, WN, RM, J1, false); \
    } while (0)
#define TAP8_LAUNCH(WGM, WGN, WMT, WN)                                                                                   \
    do {                                                                                                                \
        using Cfg8 = Tap8Cfg<WGM, WGN, WMT, WN>;                                                                        \
        p.mtiles = cdiv(p.M, Cfg8::BM);                                                                                 \
        p.ntiles = p.N / Cfg8::BN;                                                                                      \
        const long long blocks = (long long)p.B * p.mtiles * p.ntiles;                                                  \
        ProfScope ps(h, st, (std::string("tap_gemm8_kernel<" #WGM ", " #WGN ", " #WMT ", " #WN ", 2>") + shape).c_str(), flops, bytes); \
        if (s0.J == 1) { if (p.amax_rows) TAP8_LAUNCH_K(WGM, WGN, WMT, WN, true, true); else TAP8_LAUNCH_K(WGM, WGN, WMT, WN, false, true); } \
        else           { TAP8_LAUNCH_K(WGM, WGN, WMT, WN, false, false); }   /* (row mode is a one-tap affair: run_tap's `rowmode`) */ \
    } while (0)
                    if (form == 1) TAP8_LAUNCH(2, 4, 4, 2);
                    else if (form == 3) TAP8_LAUNCH(2, 4, 2, 2);
                    else TAP8_LAUNCH(4, 2, 2, 2);
#undef TAP8_LAUNCH_K
#undef TAP8_LAUNCH_KP
#undef TAP8_LAUNCH
                    HIPCHK(h, hipGetLastError());
                    return AC_OK;
                }
            }
        }
        // Tile / wave arrangement (measured, profiles/r2_tapgemm_variants.md).  The weight fragments come L2 -> registers and the
        // activation slab is shared through LDS, so the CU's vector-memory path and the LDS pipe are what an arrangement must
        // spare:  1 x 4 waves of 128 x 32 (distinct weight fragments per wave) beats 2 x 2 waves of 64 x 64 by 5-7 %;
        // 1 x 4 waves of 128 x 64 over 256 columns (half the A-slab reads, loads and splits per MFMA; lean main loop) gains
        // another 8-10 % where the launch still fills the chip evenly; 1 x 8 waves over 256 columns (one workgroup per CU)
        // wins for long contractions.  Choice by a small cost model: rate of the arrangement x how evenly its workgroups
        // fill the 256 CUs (waves of workgroups / ceil(waves)).
        int pick = 0;   // 0: 128 columns, 1: 256 columns lean, 2: 256 columns 1 x 8
        if (p.N % 256 == 0) {
            const double wg256 = (double)p.B * cdiv(p.M, 128) * (p.N / 256);
            auto fill = [](double wgs, double slots) { const double w = wgs / slots; return w / std::ceil(w); };
            // (split16: the 128-column arrangement runs three workgroups per CU and is 6 % faster per flop than before)
            const double s128 = p.winv ? 1.06 * fill(2.0 * wg256, 768.0) : 1.00 * fill(2.0 * wg256, 512.0);
            const double s256 = 1.10 * fill(wg256, 512.0);
            // (split16: 1 x 8 waves no longer beat the three-workgroup 128-column arrangement per flop -- WavTokenizer's K = 2304 layers:
            //  4.71 ms at 128 columns, 5.43 ms with 1 x 8 waves)
            const double s8 = (kk >= 2048 ? (p.winv ? 1.00 : 1.12) : (p.winv ? 0.85 : 0.95)) * fill(wg256, 256.0);
            pick = s256 >= s128 && s256 >= s8 ? 1 : (s8 > s128 ? 2 : 0);
            if (h->dev.tap_pick >= 0 && h->dev.tap_pick <= 2) pick = h->dev.tap_pick;     // developer override
        }
        // (256-row, 8-wave arrangements of THIS kernel -- <2,4,4,2>, <2,4,4,1> -- measured 7-12 % / 25-30 % slower per layer than
        //  the picks below: one workgroup per CU and the old load pipeline; profiles/r4_tapgemm8.md.  tap_gemm8.h is that tile with a
        //  pipeline built for it.)
        if (pick == 1) TAP6_CASE(1, 4, 4, 2);
        else if (pick == 2) TAP6_CASE(1, 8, 4, 1);
        else if (p.N % 128 == 0) TAP6_CASE(1, 4, 4, 1);
        else if (p.N % 192 == 0) TAP6_CASE(2, 2, 2, 3);   // DAC's 192-wide layers: a weight fragment is loaded by two waves, not four
        else if (p.N % 96 == 0) TAP6_CASE(4, 1, 1, 3);
        else TAP6_CASE(2, 2, 2, 1);
#undef TAP6_CASE
#undef TAP6_LAUNCH
        HIPCHK(h, hipGetLastError());
        return AC_OK;
    }
    if (p.N <= 16) TAP_CASE(4, 1, 2, 1);
    else if (p.N <= 32) TAP_CASE(4, 1, 2, 2);
    else if (p.N <= 64) TAP_CASE(2, 2, 2, 2);
    else if (p.N % 96 == 0 && p.N % 128 != 0) TAP_CASE(2, 2, 4, 3);   // DAC widths 96 / 192: 128-column tiles would idle a quarter of the MFMAs
    else TAP_CASE(2, 2, 4, 4);
#undef TAP_CASE
    if (rc) return rc;
    HIPCHK(h, hipGetLastError());
    return AC_OK;
}


// the [64][128] layers on 64-float super-rows (thin_conv6.h); returns 1 when the shape does not qualify
int try_thin6(ac_handle* h, hipStream_t st, const PackedGemm& g, const Act& x, int width, int edge, Out out, int B, const unsigned** amax_out) {
    if (h->gemm_fp32 || g.N != 64 || g.Ktot != 128 || x.C != width || x.ts != width || x.bs != (long long)x.L * width ||
        (x.L * width) % 64 || x.L * width < 256 || !aligned16(x.p) || (long long)x.L * width * 4 > 0x70000000LL)
        return 1;
    auto it = h->t6_of.find(g.w_off);
    if (it == h->t6_of.end()) return 1;
    ThinConv6Params p{};
    p.x = x.p;
    p.wf = reinterpret_cast<const __bf16*>(h->blob + it->second);
    p.bias = h->blob + g.b_off;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.B = B;
    p.Ls = x.L * width / 64;
    p.M = p.Ls;
    p.ntiles = cdiv(p.M, T6_BM);
    p.edge = edge;
    const bool s16 = h->t6inv_of.count(g.w_off) != 0;
    if (s16) {
        p.amax_in = amax_of(h, st, x.p, x.bs, x.ts, x.L, x.C, B, x.amax_n == B ? x.amax : nullptr);
        if (!p.amax_in) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
        p.winv = h->blob + h->t6inv_of[g.w_off];
        p.amax_out = amax_new(h);
        if (amax_out) *amax_out = p.amax_out;
    }
    const long long total = (long long)B * p.ntiles;
    const int grid = (int)std::min<long long>(total, 4 * 256);   // persistent, four workgroups per CU
    ProfScope ps(h, st, "thin_conv6_kernel<2>", 2.0 * B * p.M * 64.0 * 128.0,
                 (double)B * p.M * 256.0 * (1 + (out.raw ? 1 : 0) + (out.elu ? 1 : 0)));
    hipLaunchKernelGGL(thin_conv6_kernel<2>, dim3(grid), dim3(256), T6_LDS, st, p);
    return AC_OK;
}


// conv (stride 1 or k = 2*stride), causal reflect padding.
int conv_fwd(ac_handle* h, hipStream_t st, const PackedGemm& g, const Act& x, int k, int s, const float* rel_len, Out out,
             long long out_bs, long long out_rs, int B, Act2* y) {
    const int M = cdiv(x.L, s);
    const int extra = M * s - x.L;
    if (s == 2 && k == 4 && !rel_len && extra == 0 && x.L >= 4 && out_rs == 64 && out_bs == (long long)M * 64 && !h->noncausal) {
        const unsigned* am = nullptr;
        const int rc = try_thin6(h, st, g, x, 32, 1, out, B, &am);
        if (rc <= 0) {
            if (y && !rc) {
                y->raw = Act{out.raw, out_bs, out_rs, M, g.N, am, B};
                y->elu = Act{out.elu, out_bs, out_rs, M, g.N, am, B};
            }
            if (!rc) HIPCHK(h, hipGetLastError());
            return rc;
        }
    }
    TapGemmParams p{};
    p.nseg = 1;
    if (s != 1 && k != 2 * s) return fail(h, AC_EINVAL, "strided conv needs kernel == 2*stride (got k=%d, s=%d)", k, s);
    if (h->noncausal) {   // padding_total = k - s; right = total / 2, left = total - right
        const int total = k - s, right = total / 2;
        p.seg[0] = make_seg(x, s, s == 1 ? k : 2, PAD_REFLECT, extra, 0, rel_len, total - right, right);
    } else {
        p.seg[0] = make_seg(x, s, s == 1 ? k : 2, PAD_REFLECT, extra, 0, rel_len);
    }
    p.w = h->blob + g.w_off;
    p.bias = g.has_bias ? h->blob + g.b_off : nullptr;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.y_bs = out_bs;
    p.y_rs = out_rs;
    p.B = B;
    p.M = M;
    p.N = g.N;
    p.Ktot = g.Ktot;
    const int rc = run_tap(h, st, p);
    if (y) {
        y->raw = Act{out.raw, out_bs, out_rs, M, g.N, p.amax_out, p.B};
        y->elu = Act{out.elu, out_bs, out_rs, M, g.N, p.amax_out, p.B};
    }
    return rc;
}


int convtr_fwd(ac_handle* h, hipStream_t st, const PackedGemm& g, const Act& x, int s, Out out, int B, Act2* y) {
    const int cout = g.N / s;
    if (s == 2) {
        const unsigned* am = nullptr;
        const int rc = try_thin6(h, st, g, x, 64, 0, out, B, &am);
        if (rc <= 0) {
            if (!rc) {
                y->raw = Act{out.raw, (long long)x.L * s * cout, cout, x.L * s, cout, am, B};
                y->elu = Act{out.elu, (long long)x.L * s * cout, cout, x.L * s, cout, am, B};
                HIPCHK(h, hipGetLastError());
            }
            return rc;
        }
    }
    TapGemmParams p{};
    p.nseg = 1;
    p.seg[0] = make_seg(x, 1, 2, PAD_ZERO, 0, 0, nullptr);
    p.w = h->blob + g.w_off;
    p.bias = h->blob + g.b_off;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.y_bs = (long long)x.L * g.N;
    p.y_rs = g.N;
    p.B = B;
    p.M = x.L;
    p.N = g.N;
    p.Ktot = g.Ktot;
    const int rc = run_tap(h, st, p);
    y->raw = Act{out.raw, (long long)x.L * s * cout, cout, x.L * s, cout, p.amax_out, p.B};
    y->elu = Act{out.elu, (long long)x.L * s * cout, cout, x.L * s, cout, p.amax_out, p.B};
    return rc;
}


template <int C, int BM, int NSPLIT, bool SC = true>
int launch_rb_fused(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, Out out, int B, int pad = PAD_REFLECT) {
    using Cfg = RbCfg<C, BM, NSPLIT, SC>;
    RbFusedParams p{};
    p.xe = x.elu.p;   // may be null: the kernel then activates the raw rows itself
    p.xr = x.raw.p;
    p.w3 = h->blob + rb.c3.w_off;
    p.b3 = h->blob + rb.c3.b_off;
    p.wf = h->blob + rb.fused.w_off;
    p.bf = h->blob + rb.fused.b_off;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.B = B;
    p.L = x.raw.L;
    p.Lp = x.raw.L > 2 ? x.raw.L : 3;
    p.ntiles = cdiv(x.raw.L, BM);
    p.pad = pad;
    if (int rc = ensure_lds(h, reinterpret_cast<const void*>(rb_fused_kernel<C, BM, NSPLIT, SC>), Cfg::lds_bytes)) return rc;
    const long long total = (long long)B * p.ntiles;
    const int grid = (int)std::min<long long>(total, 512);   // two workgroups per CU (VGPR-limited), persistent
    const size_t lds = Cfg::lds_bytes;
    const double L = x.raw.L;
    ProfScope ps(h, st, !SC ? "rb_fused_kernel<64, 64, 2, false>" : C == 32 ? "rb_fused_kernel<32, 128, 1>" : "rb_fused_kernel<64, 64, 2>",
                 2.0 * B * L * ((double)(C / 2) * 3 * C + (double)C * Cfg::KF),
                 (double)B * L * C * 4.0 * ((x.elu.p ? 2 : 1) + (out.raw ? 1 : 0) + (out.elu ? 1 : 0)));
    hipLaunchKernelGGL((rb_fused_kernel<C, BM, NSPLIT, SC>), dim3(grid), dim3(256), lds, st, p);
    return AC_OK;
}


// split-operand version of the fused block (rb_fused6.h); reads the raw rows only and activates them itself
// split16.h operands of a fused block launch; false when the block's images are bf16 planes
bool rb_split16(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, int B, RbFused6Params& p, const unsigned** amax_out) {
    if (h->gemm_fp32 || !rb.winv3_off) return false;
    p.amax_in = amax_of(h, st, x.raw.p, x.raw.bs, x.raw.ts, x.raw.L, x.raw.C, B, x.raw.amax_n == B ? x.raw.amax : nullptr);
    p.winv3 = h->blob + rb.winv3_off;
    p.winvf = h->blob + rb.winvf_off;
    p.hb0 = rb.hb0;
    p.hb1 = rb.hb1;
    p.amax_out = amax_new(h);
    if (amax_out) *amax_out = p.amax_out;
    return true;
}


template <int C, bool SC>
int launch_rb_fused6(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, Out out, int B, int pad = PAD_REFLECT, const unsigned** amax_out = nullptr,
                     const PackedGemm* head = nullptr, int head_k = 0, float* head_y = nullptr) {
    using Cfg = Rb6Cfg<C, SC>;
    RbFused6Params p{};
    const bool with_head = head && C == 64 && !SC;
    if (with_head) {            // the decoder's final Conv1d(C, 1, k) applied to the block's output tile in LDS (rb_fused6.h HEAD)
        p.head_w = h->blob + head->w_off;
        p.head_b = h->blob + head->b_off;
        p.head_y = head_y;
        p.head_k = head_k;
    }
    p.xr = x.raw.p;
    p.w3f = reinterpret_cast<const __bf16*>(h->blob + rb.w3f_off);
    p.wff = reinterpret_cast<const __bf16*>(h->blob + rb.wff_off);
    p.b3 = h->blob + rb.c3.b_off;
    p.bf = h->blob + rb.fused.b_off;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.B = B;
    p.L = x.raw.L;
    p.lpad = h->noncausal ? 1 : 2;
    p.Lp = x.raw.L > p.lpad ? x.raw.L : p.lpad + 1;
    p.ntiles = cdiv(x.raw.L, with_head ? Cfg::BM - (head_k - 1) : Cfg::BM);
    p.pad = pad;
    p.dbg = h->dev.rb6_dbg;
    if (!rb_split16(h, st, rb, x, B, p, amax_out)) return fail(h, AC_ESTATE, "fused residual block without split16 images");
    if (!p.amax_in) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
    if (with_head) p.amax_out = nullptr;        // nothing reads the block's output but the head
    const double L = x.raw.L;
    if (C == 64 && !with_head && h->dev.rb_stream && p.lpad == 2 && rb.w3p_off && (long long)x.raw.L * 256 < 0x70000000LL)
        return launch_rb_stream6(h, st, p, rb, SC, out, B);      // rb_stream6.h (stream_path.hip): 16 waves per CU, one wave = one stream
    const size_t lds6 = with_head ? Cfg::lds_bytes16_head : Cfg::lds_bytes16;
    if (int rc = ensure_lds(h, with_head ? reinterpret_cast<const void*>(rb_fused6_head_kernel) : reinterpret_cast<const void*>(rb_fused6_kernel<C, SC, 2>), lds6)) return rc;
    const long long total = (long long)B * p.ntiles;
    const int per_cu = with_head ? RB6_HEAD_OCC : rb6_occupancy<C, SC, 2>();
    const int grid = (int)std::min<long long>(total, (long long)per_cu * 256);   // persistent
    ProfScope ps(h, st, with_head ? "rb_fused6_head_kernel" : !SC ? "rb_fused6_kernel<64, false, 2>" : C == 32 ? "rb_fused6_kernel<32, true, 2>" : "rb_fused6_kernel<64, true, 2>",
                 2.0 * B * L * ((double)(C / 2) * 3 * C + (double)C * (C / 2 + (SC ? C : 0)) + (with_head ? (double)head_k * C : 0.0)),
                 with_head ? (double)B * L * (C + 1) * 4.0 : (double)B * L * C * 4.0 * (1 + (out.raw ? 1 : 0) + (out.elu ? 1 : 0)));
    if (with_head) hipLaunchKernelGGL(rb_fused6_head_kernel, dim3(grid), dim3(256), lds6, st, p);
    else hipLaunchKernelGGL((rb_fused6_kernel<C, SC, 2>), dim3(grid), dim3(256), lds6, st, p);
    return AC_OK;
}


// the 128-channel block as one 8-wave workgroup per CU (rb_fused6_128.h)
template <bool SC>
int launch_rb128_fused6(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, Out out, int B, int pad = PAD_REFLECT, const unsigned** amax_out = nullptr) {
    using Cfg = Rb128Cfg<SC>;
    RbFused6Params p{};
    p.xr = x.raw.p;
    p.w3f = reinterpret_cast<const __bf16*>(h->blob + rb.w3f_off);
    p.wff = reinterpret_cast<const __bf16*>(h->blob + rb.wff_off);
    p.b3 = h->blob + rb.c3.b_off;
    p.bf = h->blob + rb.fused.b_off;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.B = B;
    p.L = x.raw.L;
    p.lpad = h->noncausal ? 1 : 2;
    p.Lp = x.raw.L > p.lpad ? x.raw.L : p.lpad + 1;
    p.ntiles = cdiv(x.raw.L, Cfg::BM);
    p.pad = pad;
    if (!rb_split16(h, st, rb, x, B, p, amax_out)) return fail(h, AC_ESTATE, "fused residual block without split16 images");
    if (!p.amax_in) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
    if (h->dev.rb_stream && h->dev.rb128_stream && p.lpad == 2 && pad == (SC ? PAD_REFLECT : PAD_ZERO) && rb.w3p_off && (long long)x.raw.L * 512 < 0x70000000LL)
        return launch_rb_stream128m(h, st, p, rb, SC, out, B);  // rb_stream128m.h (stream_path.hip): twelve / sixteen waves per CU, no slab
    if (int rc = ensure_lds(h, reinterpret_cast<const void*>(rb128_fused6_kernel<SC, 2>), Cfg::lds_bytes)) return rc;
    const long long total = (long long)B * p.ntiles;
    const int grid = (int)std::min<long long>(total, 256);   // persistent, one workgroup per CU
    const double L = x.raw.L;
    ProfScope ps(h, st, SC ? "rb128_fused6_kernel<true, 2>" : "rb128_fused6_kernel<false, 2>",
                 2.0 * B * L * (64.0 * 384 + 128.0 * (64 + (SC ? 128 : 0))),
                 (double)B * L * 128 * 4.0 * (1 + (out.raw ? 1 : 0) + (out.elu ? 1 : 0)));
    hipLaunchKernelGGL((rb128_fused6_kernel<SC, 2>), dim3(grid), dim3(512), Cfg::lds_bytes, st, p);
    return AC_OK;
}

// can the 128-channel block run fused?  (the producer then writes the raw flavour only)
bool rb128_ok(const ac_handle* h, const ResBlockPlan& rb) { return rb.C == 128 && rb.has6 && !h->gemm_fp32; }


// ResBlock: hbuf = ELU(conv3(ELU(x)));  out = [hbuf | x] * [W1; Ws] + (b1 + bs)
int resblock_fwd(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, float* hbuf, Out out, int B, Act2* y) {
    if (rb128_ok(h, rb) && h->cfg.residual_kernel_size == 3 && h->cfg.compress == 2 && x.raw.ts == 128 &&
        x.raw.bs == (long long)x.raw.L * 128 && aligned16(x.raw.p) && (long long)x.raw.L * 512 < 0x70000000LL) {
        const unsigned* am = nullptr;
        int rc = launch_rb128_fused6<true>(h, st, rb, x, out, B, PAD_REFLECT, &am);
        if (rc) return rc;
        HIPCHK(h, hipGetLastError());
        const long long bs = (long long)x.raw.L * 128;
        y->raw = Act{out.raw, bs, 128, x.raw.L, 128, am, B};
        y->elu = Act{out.elu, bs, 128, x.raw.L, 128, am, B};
        return AC_OK;
    }
    // thin stages: one fused kernel, hidden activation never leaves the CU
    // (rb_fused.h, the exact-product version, knows the causal halo only)
    if ((rb.C == 32 || rb.C == 64) && !(h->noncausal && (!rb.has6 || h->gemm_fp32)) && h->cfg.residual_kernel_size == 3 && h->cfg.compress == 2 && x.raw.ts == rb.C &&
        x.raw.bs == (long long)x.raw.L * rb.C && aligned16(x.raw.p) &&
        (!x.elu.p || (x.elu.ts == rb.C && x.elu.bs == x.raw.bs && aligned16(x.elu.p)))) {
        int rc;
        const unsigned* am = nullptr;
        if (rb.has6 && !h->gemm_fp32) rc = rb.C == 32 ? launch_rb_fused6<32, true>(h, st, rb, x, out, B, PAD_REFLECT, &am) : launch_rb_fused6<64, true>(h, st, rb, x, out, B, PAD_REFLECT, &am);
        else rc = rb.C == 32 ? launch_rb_fused<32, 128, 1>(h, st, rb, x, out, B) : launch_rb_fused<64, 64, 2>(h, st, rb, x, out, B);   // <C, rows per tile, column split>
        if (rc) return rc;
        HIPCHK(h, hipGetLastError());
        const long long bs = (long long)x.raw.L * rb.C;
        y->raw = Act{out.raw, bs, rb.C, x.raw.L, rb.C, am, B};
        y->elu = Act{out.elu, bs, rb.C, x.raw.L, rb.C, am, B};
        return AC_OK;
    }
    Act2 hv;
    int rc = conv_fwd(h, st, rb.c3, x.elu, h->cfg.residual_kernel_size, 1, nullptr, Out{nullptr, hbuf},
                      (long long)x.elu.L * rb.c3.N, rb.c3.N, B, &hv);
    if (rc) return rc;
    TapGemmParams p{};
    p.nseg = 2;
    p.seg[0] = make_seg(hv.elu, 1, 1, PAD_REFLECT, 0, 0, nullptr);
    p.seg[1] = make_seg(x.raw, 1, 1, PAD_REFLECT, 0, hv.elu.C, nullptr);
    p.w = h->blob + rb.fused.w_off;
    p.bias = h->blob + rb.fused.b_off;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.y_bs = (long long)x.raw.L * rb.C;
    p.y_rs = rb.C;
    p.B = B;
    p.M = x.raw.L;
    p.N = rb.C;
    p.Ktot = rb.fused.Ktot;
    rc = run_tap(h, st, p);
    y->raw = Act{out.raw, p.y_bs, p.y_rs, x.raw.L, rb.C, p.amax_out, p.B};
    y->elu = Act{out.elu, p.y_bs, p.y_rs, x.raw.L, rb.C, p.amax_out, p.B};
    return rc;
}


// stem / head: dedicated HBM-bound kernels when the shape allows, tap-GEMM otherwise
bool thin_ok(const ac_config& c, int k) { return c.num_filters % 4 == 0 && c.num_filters <= 64 && k <= THIN_MAXK; }


int thin_stem(ac_handle* h, hipStream_t st, const PackedGemm& g, int F, int k, int pad, const float* sig, const float* rel_len, int B, int T,
              Out out, Act2* y, int padl, const float* alpha, const float* alpha_inv, int Lp) {
    ThinParams p{};
    p.padl = padl < 0 ? k - 1 : padl;
    p.alpha = alpha;
    p.alpha_inv = alpha_inv;
    p.x = sig;
    p.w = h->blob + g.w_off;
    p.bias = h->blob + g.b_off;
    p.rel_len = rel_len;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.B = B;
    p.T = T;
    p.F = F;
    p.k = k;
    p.Lp = Lp > 0 ? Lp : (T > k - 1 ? T : k);
    p.pad = pad;
    p.amax_out = amax_new(h);
    {
        ProfScope ps(h, st, "stem_kernel", 2.0 * B * (double)T * F * k,
                     (double)B * T * 4.0 * (1 + F * ((out.raw ? 1 : 0) + (out.elu ? 1 : 0))));
        hipLaunchKernelGGL(stem_kernel, dim3(cdiv(T, STEM_TT), B), dim3(256), 0, st, p);
    }
    HIPCHK(h, hipGetLastError());
    y->raw = Act{out.raw, (long long)T * F, F, T, F, p.amax_out, B};
    y->elu = Act{out.elu, (long long)T * F, F, T, F, p.amax_out, B};
    return AC_OK;
}


int stem_fwd(ac_handle* h, hipStream_t st, const float* sig, const float* rel_len, int B, int T, Out out, Act2* y) {
    const int k = h->cfg.kernel_size;
    if (h->noncausal) {   // centred: right = (k-1)/2, left = k-1-right; small-input rule on max(left, right)
        const int right = (k - 1) / 2, left = k - 1 - right;
        return thin_stem(h, st, h->enc_stem, h->cfg.num_filters, k, PAD_REFLECT, sig, rel_len, B, T, out, y, left, nullptr, nullptr, T > left ? T : left + 1);
    }
    return thin_stem(h, st, h->enc_stem, h->cfg.num_filters, k, PAD_REFLECT, sig, rel_len, B, T, out, y);
}


int thin_head(ac_handle* h, hipStream_t st, const PackedGemm& g, int F, int k, int pad, const Act& x, int B, float* sig, int padl,
              int tanh_out) {
    ThinParams p{};
    p.padl = padl < 0 ? k - 1 : padl;
    p.tanh_out = tanh_out;
    p.x = x.p;
    p.w = h->blob + g.w_off;
    p.bias = h->blob + g.b_off;
    p.y = sig;
    p.B = B;
    p.T = x.L;
    p.F = F;
    p.k = k;
    p.Lp = x.L > k - 1 ? x.L : k;
    p.pad = pad;
    if (F % 16 == 0 && F <= 128 && k <= THIN_MAXK && !h->dev.head_seq) {      // four lanes per sample (thin.h head4_kernel; its slab is sized for F <= 128, k <= THIN_MAXK like head_kernel's)
        const size_t lds4 = ((size_t)(HEAD4_TT + THIN_MAXK) * (F + 4) + (size_t)THIN_MAXK * F) * sizeof(float);
        if (int rc = ensure_lds(h, reinterpret_cast<const void*>(head4_kernel), lds4)) return rc;
        ProfScope ps(h, st, "head4_kernel", 2.0 * B * (double)x.L * F * k, (double)B * x.L * 4.0 * (F + 1));
        hipLaunchKernelGGL(head4_kernel, dim3(cdiv(x.L, HEAD4_TT), B), dim3(256), lds4, st, p);
        HIPCHK(h, hipGetLastError());
        return AC_OK;
    }
    const size_t lds = ((size_t)(HEAD_TT + THIN_MAXK) * (F + 4) + (size_t)THIN_MAXK * F) * sizeof(float);
    if (int rc = ensure_lds(h, reinterpret_cast<const void*>(head_kernel), lds)) return rc;
    {
        ProfScope ps(h, st, "head_kernel", 2.0 * B * (double)x.L * F * k, (double)B * x.L * 4.0 * (F + 1));
        hipLaunchKernelGGL(head_kernel, dim3(cdiv(x.L, HEAD_TT), B), dim3(256), lds, st, p);
    }
    HIPCHK(h, hipGetLastError());
    return AC_OK;
}


int head_fwd(ac_handle* h, hipStream_t st, const Act& x, int B, float* sig) {
    return thin_head(h, st, h->dec_head, h->cfg.num_filters, h->cfg.last_kernel_size, PAD_REFLECT, x, B, sig);
}


// ---- fused thin-channel head of the encoder (enc_front.h): stem -> ResBlock(32) -> ELU -> Conv1d(32, 64, k4, s2)
bool enc_front_ok(const ac_handle* h, int T) {
    const ac_config& c = h->cfg;
    return h->fuse_chains && h->arch == ARCH_ENCODEC && !h->noncausal && !h->gemm_fp32 && h->enc_front.ok &&
           c.num_filters == 32 && c.kernel_size == 7 && c.residual_kernel_size == 3 && c.compress == 2 && c.num_ratios >= 1 &&
           c.upsampling_ratios[c.num_ratios - 1] == 2 && T >= 64 && (long long)T * 128 < 0x70000000LL;
}


int enc_front_fwd(ac_handle* h, hipStream_t st, const float* sig, const float* rel_len, int B, int T, float* y, float* dbg_x0, float* dbg_y1, Act2* out) {
    const ResBlockPlan& rb = h->enc_rb[0];
    const PackedGemm& gd = h->enc_down[0];
    EncFrontParams p{};
    p.sig = sig;
    p.rel_len = rel_len;
    p.w0 = h->blob + h->enc_stem.w_off;
    p.b0 = h->blob + h->enc_stem.b_off;
    p.w3f = reinterpret_cast<const __bf16*>(h->blob + rb.w3f_off);
    p.wff = reinterpret_cast<const __bf16*>(h->blob + rb.wff_off);
    p.wdf = reinterpret_cast<const __bf16*>(h->blob + h->t6_of[gd.w_off]);
    p.b3 = h->blob + rb.c3.b_off;
    p.winv3 = h->blob + rb.winv3_off;
    p.bf = h->blob + rb.fused.b_off;
    p.winvf = h->blob + rb.winvf_off;
    p.bd = h->blob + gd.b_off;
    p.winvd = h->blob + h->t6inv_of[gd.w_off];
    p.y = y;
    p.dbg_x0 = dbg_x0;
    p.dbg_y1 = dbg_y1;
    p.B = B;
    p.T = T;
    p.M = cdiv(T, 2);
    const int nchunks = cdiv(T, EF_ROWS);
    p.seg_chunks = std::max(8, cdiv(nchunks, std::max(1, 6144 / B)));     // ~6144 streams: three rounds of 2048 resident waves
    if (h->dev.front_seg > 0) p.seg_chunks = h->dev.front_seg;
    p.segs_per_clip = cdiv(nchunks, p.seg_chunks);
    // amax of the samples (one read of 4 B per sample; as 16-byte vectors where the clip pitch allows)
    const bool v4 = T % 4 == 0 && aligned16(sig);
    p.amax_sig = amax_of(h, st, sig, T, v4 ? 4 : 1, v4 ? T / 4 : T, v4 ? 4 : 1, B, nullptr);
    if (!p.amax_sig) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
    p.amax_out = amax_new(h);
    p.sb0 = h->enc_front.sb0; p.sb1 = h->enc_front.sb1;
    p.hb0 = rb.hb0; p.hb1 = rb.hb1;
    p.fb0 = h->enc_front.fb0; p.fb1h = h->enc_front.fb1h; p.fb1x = h->enc_front.fb1x;
    if (h->dev.chain_stream && h->simg.enc_ok) {       // round 6: the sixteen-waves-per-CU form (enc_stream.h, stream_path.hip)
        if (int rc = enc_stream_fwd(h, st, sig, rel_len, B, T, y, dbg_x0, dbg_y1, p.amax_sig, p.amax_out)) return rc;
        HIPCHK(h, hipGetLastError());
        out->raw = Act{y, (long long)p.M * 64, 64, p.M, 64, p.amax_out, B};
        out->elu = Act{nullptr, (long long)p.M * 64, 64, p.M, 64, p.amax_out, B};
        return AC_OK;
    }
    size_t lds = EF_LDS;
    lds += (size_t)h->dev.front_ldspad;     // developer: force one workgroup per CU
    if (int rc = ensure_lds(h, reinterpret_cast<const void*>(enc_front_kernel), lds)) return rc;
    const long long streams = (long long)B * p.segs_per_clip;
    {
        ProfScope ps(h, st, "enc_front_kernel", 2.0 * B * (double)T * (7.0 * 32 + 16.0 * 96 + 32.0 * 48 + 64.0 * 128 / 2),
                     (double)B * T * 4.0 + (double)B * p.M * 256.0);
        hipLaunchKernelGGL(enc_front_kernel, dim3((unsigned)cdiv((int)streams, EF_WAVES)), dim3(64 * EF_WAVES), lds, st, p);
    }
    HIPCHK(h, hipGetLastError());
    out->raw = Act{y, (long long)p.M * 64, 64, p.M, 64, p.amax_out, B};
    out->elu = Act{nullptr, (long long)p.M * 64, 64, p.M, 64, p.amax_out, B};
    return AC_OK;
}


// ---- fused thin-channel tail of the decoder (dec_tail.h): ConvTranspose1d(64, 32, k4, s2) -> ResBlock(32) -> ELU -> Conv1d(32, 1, k7)
bool dec_tail_ok(const ac_handle* h, const Act& xe) {
    const ac_config& c = h->cfg;
    return h->fuse_chains && h->arch == ARCH_ENCODEC && !h->noncausal && !h->gemm_fp32 && h->dec_tail.ok &&
           c.num_filters == 32 && c.last_kernel_size == 7 && c.residual_kernel_size == 3 && c.compress == 2 && c.num_ratios >= 1 &&
           c.upsampling_ratios[c.num_ratios - 1] == 2 && xe.p && xe.C == 64 && xe.ts == 64 && xe.bs == (long long)xe.L * 64 && aligned16(xe.p) &&
           xe.L >= 32 && (long long)xe.L * 256 < 0x70000000LL;
}


int dec_tail_fwd(ac_handle* h, hipStream_t st, const Act& xe, int B, float* sig, float* dbg_u, float* dbg_v) {
    const int last = h->cfg.num_ratios - 1;
    const ResBlockPlan& rb = h->dec_rb[last];
    const PackedGemm& gu = h->dec_up[last];
    DecTailParams p{};
    p.xe = xe.p;
    p.wuf = reinterpret_cast<const __bf16*>(h->blob + h->t6_of[gu.w_off]);
    p.w3f = reinterpret_cast<const __bf16*>(h->blob + rb.w3f_off);
    p.wff = reinterpret_cast<const __bf16*>(h->blob + rb.wff_off);
    p.bu = h->blob + gu.b_off;
    p.winvu = h->blob + h->t6inv_of[gu.w_off];
    p.b3 = h->blob + rb.c3.b_off;
    p.winv3 = h->blob + rb.winv3_off;
    p.bf = h->blob + rb.fused.b_off;
    p.winvf = h->blob + rb.winvf_off;
    p.wh = h->blob + h->dec_head.w_off;
    p.bh = h->blob + h->dec_head.b_off;
    p.sig = sig;
    p.dbg_u = dbg_u;
    p.dbg_v = dbg_v;
    p.B = B;
    p.L = xe.L;
    const int nchunks = cdiv(xe.L, DT_ROWS);
    p.seg_chunks = std::max(8, cdiv(nchunks, std::max(1, 6144 / B)));
    if (h->dev.tail_seg > 0) p.seg_chunks = h->dev.tail_seg;
    p.segs_per_clip = cdiv(nchunks, p.seg_chunks);
    p.amax_x = amax_of(h, st, xe.p, xe.bs, xe.ts, xe.L, xe.C, B, xe.amax_n == B ? xe.amax : nullptr);
    if (!p.amax_x) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
    p.ub0 = h->dec_tail.sb0; p.ub1 = h->dec_tail.sb1;
    p.hb0 = rb.hb0; p.hb1 = rb.hb1;
    if (h->dev.chain_stream && h->simg.dec_ok) {       // round 6: the sixteen-waves-per-CU form (dec_stream.h, stream_path.hip)
        if (int rc = dec_stream_fwd(h, st, xe, B, sig, dbg_u, dbg_v, p.amax_x)) return rc;
        HIPCHK(h, hipGetLastError());
        return AC_OK;
    }
    if (int rc = ensure_lds(h, reinterpret_cast<const void*>(dec_tail_kernel), DT_LDS)) return rc;
    const long long streams = (long long)B * p.segs_per_clip;
    {
        ProfScope ps(h, st, "dec_tail_kernel", 2.0 * B * (double)xe.L * (64.0 * 128 + 2.0 * (16.0 * 96 + 32.0 * 48 + 7.0 * 32)),
                     (double)B * xe.L * 256.0 + (double)B * xe.L * 8.0);
        hipLaunchKernelGGL(dec_tail_kernel, dim3((unsigned)cdiv((int)streams, 8)), dim3(512), DT_LDS, st, p);
    }
    HIPCHK(h, hipGetLastError());
    return AC_OK;
}


void capture(ac_handle* h, hipStream_t st, const Act& a, int B) {
    if (!h->dbg) return;
    const size_t n = (size_t)B * a.L * a.C;
    if (h->dbg_used + n <= h->dbg_cap)
        (void)hipMemcpyAsync(h->dbg + h->dbg_used, a.p, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    h->dbg_used += n;
}


// x [B][T][D] (standard layout) -> lstm(x) + x as raw and/or ELU'd [B][T][D]
int lstm_fwd(ac_handle* h, hipStream_t st, const LstmPlan& lp, const Act& x, const LstmWs& ws, Out out, int B, Act2* y) {
    const int D = lp.D, T = x.L, L = lp.layers;
    if (D % 64 != 0 || D > 512) return fail(h, AC_EINVAL, "LSTM width %d unsupported (need 64, 128, 256 or 512)", D);
    if (L < 1 || L > 2) return fail(h, AC_EINVAL, "%d LSTM layers unsupported (1 or 2)", L);
    // Under stream capture the per-step kernels run: the persistent kernel assigns roles from the XCD its workgroups land on and
    // needs all 256 of them co-resident from the start; a cooperative launch guarantees that, its replay from a hipGraph does not
    // (observed: 7 of 8 replayed launches without 32 workgroups on every XCD -- detected by the kernel, outputs NaN, status raised).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    const bool persist = lp.has_persist && h->lp_ctl && h->num_cus == 256 && !h->lstm_step_only && x.ts == D && cap == hipStreamCaptureStatusNone;
    const bool fuse_env = h->dev.lstm_fuse_in != 0;
    const bool fuse_in = persist && !h->gemm_fp32 && fuse_env && aligned16(x.p) && x.bs % 4 == 0;   // lstm_persist16.h computes W_ih0 * x[t] itself
    // layer-0 input projection for all t: gin[t][b][4D]
    if (!fuse_in) {
        TapGemmParams p{};
        p.nseg = 1;
        p.seg[0] = make_seg(x, 1, 1, PAD_ZERO, 0, 0, nullptr);
        p.w = h->blob + lp.ih[0].w_off;
        p.bias = h->blob + lp.ih[0].b_off;
        p.y = ws.gin;
        p.y_bs = 4LL * D;
        p.y_rs = (long long)B * 4 * D;
        p.B = B;
        p.M = T;
        p.N = 4 * D;
        p.Ktot = D;
        int rc = run_tap(h, st, p);
        if (rc) return rc;
    }
    const int nroles = 2 * L - 1, nlaunch = T + 2 * (L - 1);
    const long long BD = (long long)((B + 31) / 32 * 32) * D, B4D = (long long)B * 4 * D;   // h lives in A-fragment tiles of 16 clips
    if (persist) {
        // one cooperative launch per 64 clips walks all T steps (lstm_persist.h)
        const int chunks = cdiv(B, 64);
        const unsigned* x_amax = nullptr;       // split16.h: the fused input projection scales x by its clip's amax
        if (fuse_in && lp.persist16_inv) {
            x_amax = amax_of(h, st, x.p, x.bs, x.ts, x.L, x.C, B, x.amax_n == B ? x.amax : nullptr);
            if (!x_amax) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
        }
        int* poison = reinterpret_cast<int*>(ws.c);   // the cell-state buffer of the per-step kernels is free on this path
        if (!h->gemm_fp32) HIPCHK(h, hipMemsetAsync(poison, 0x7f, (size_t)B * sizeof(int), st));
        ProfScope ps(h, st, h->gemm_fp32 ? "lstm_persist_kernel" : fuse_in ? "lstm_persist16_kernel<true>" : "lstm_persist16_kernel<false>", 2.0 * T * (double)B * 4 * D * D * (nroles + (fuse_in ? 1 : 0)),
                     (double)T * ((double)B * 4 * D * 4 + 3.0 * B * D * 4) + 12.0 * D * D * 4, chunks);
        auto tail = [&](int c0, int nb, const int* pz) {
            LstmTailParams tp{};
            tp.ctl = h->lp_ctl;
            tp.sticky = h->sticky_dev;
            tp.poison = pz;
            tp.yout = out.raw;
            tp.yout_elu = out.elu;
            tp.y_bs = (long long)T * D;
            tp.clip0 = c0;
            tp.B = nb;
            tp.T = T;
            tp.D = D;
            tp.xcds_used = 8;
            hipLaunchKernelGGL(lstm_tail_kernel, dim3(nb, 4), dim3(256), 0, st, tp);
        };
        for (int c0 = 0; c0 < B; c0 += 64) {
            LstmPersistParams q{};
            q.gin0 = ws.gin;
            q.w_pk = h->blob + lp.persist_off;
            q.bias1 = h->blob + lp.ih[1].b_off;
            q.hseq0 = ws.hseq0;
            q.hseq1 = ws.hseq1;
            q.skip = x.p;
            q.yout = out.raw;
            q.yout_elu = out.elu;
            q.ctl = h->lp_ctl;
            q.gin_ts = B4D;
            q.h_ts = BD;
            q.skip_bs = x.bs;
            q.y_bs = (long long)T * D;
            q.B = std::min(64, B - c0);
            q.T = T;
            q.group0 = c0 / 16;
            q.clip0 = c0;
            q.dbg = h->dev.lstm_dbg;
            if (!h->gemm_fp32) {   // split16 products on the fp16 pipe (lstm_persist16.h): h travels as fp16 plane blocks
                LstmPersist16Params q6{};
                q6.base = q;
                q6.base.h_ts = (long long)((B + 31) / 32 * 2) * LP16_GROUP_BYTES;
                q6.winv = h->blob + lp.persist16_inv;
                q6.amax_x = fuse_in ? x_amax : nullptr;
                q6.w_pk6 = reinterpret_cast<const __bf16*>(h->blob + lp.persist16_off);
                q6.bias0 = h->blob + lp.ih[0].b_off;
                q6.fuse_in = fuse_in ? 1 : 0;
                q6.poison = poison;
                q6.hseq0_local = ws.gin1;     // free on this path (the per-step kernels' layer-1 pre-activations)
                HIPCHK(h, hipMemsetAsync(h->lp_ctl, 0, LP_CTL_WORDS * sizeof(unsigned), st));
                // the exchange validates itself: every element of the h buffers starts as the "not yet written" pattern
                const size_t hbytes = (size_t)T * (size_t)q6.base.h_ts;
                HIPCHK(h, hipMemsetAsync(ws.hseq0, 0xFF, hbytes, st));
                HIPCHK(h, hipMemsetAsync(ws.hseq1, 0xFF, hbytes, st));
                HIPCHK(h, hipMemsetAsync(ws.gin1, 0xFF, hbytes, st));
                void* args6[] = {&q6};
                const void* kfn = fuse_in ? reinterpret_cast<const void*>(lstm_persist16_kernel<true>) : reinterpret_cast<const void*>(lstm_persist16_kernel<false>);
                HIPCHK(h, hipLaunchCooperativeKernel(kfn, dim3(256), dim3(512), args6, 0, st));
                tail(c0, q.B, poison);
                if (AC_DEV_MODE(q.dbg, 32)) {   // developer trace: 100 MHz real-time stamps of steps 100 .. 103 (lstm_persist16.h)
                    HIPCHK(h, hipStreamSynchronize(st));
                    std::vector<unsigned long long> tr(8 * 64);
                    HIPCHK(h, hipMemcpy(tr.data(), h->lp_ctl + LP_CTL_FLAGS, tr.size() * 8, hipMemcpyDeviceToHost));
                    for (int role = 0; role < 2; ++role) {
                        std::fprintf(stderr, "lstm trace role %d (layer %d), 10 ns units from step start: ", role, role & 1);
                        for (int s_ = 0; s_ < 4; ++s_) {
                            const unsigned long long* r = &tr[role * 64 + s_ * 8];
                            std::fprintf(stderr, "| t=%d:", 100 + s_);
                            for (int k = 1; k < 8; ++k) std::fprintf(stderr, " %lld", r[k] ? (long long)(r[k] - r[0]) : -1LL);
                            if (s_ < 3) std::fprintf(stderr, " next %lld ", (long long)(r[8] - r[0]));
                        }
                        const double us3 = (double)(tr[role * 64 + 24] - tr[role * 64]) * 0.01;
                        std::fprintf(stderr, " | steps that had to poll: %llu of %d | shader clock %.0f MHz\n", tr[role * 64 + 63], T,
                                     us3 > 0 ? (double)(tr[role * 64 + 33] - tr[role * 64 + 32]) / us3 : 0.0);
                    }
                }
                continue;
            }
            HIPCHK(h, hipMemsetAsync(h->lp_ctl, 0, LP_CTL_WORDS * sizeof(unsigned), st));
            void* args[] = {&q};
            HIPCHK(h, hipLaunchCooperativeKernel(reinterpret_cast<const void*>(lstm_persist_kernel), dim3(256), dim3(256), args, 0, st));
            tail(c0, q.B, nullptr);
        }
        HIPCHK(h, hipGetLastError());
        const unsigned* am = amax_plus(h, st, x, 1.0f, B);   // |lstm(x) + x| <= 1 + amax(x)
        y->raw = Act{out.raw, (long long)T * D, D, T, D, am, B};
        y->elu = Act{out.elu, (long long)T * D, D, T, D, am, B};
        return AC_OK;
    }
    {
        ProfScope ps(h, st, "lstm_step_kernel", 2.0 * T * (double)B * 4 * D * D * nroles,
                     (double)T * nroles * ((double)B * 4 * D * 4 + 4.0 * D * D * 4 + 2.0 * B * D * 4), nlaunch);
        for (int s = 0; s < nlaunch; ++s) {
            LstmLaunchParams q{};
            q.B = B;
            q.D = D;
            auto fill_last = [&](LstmRole& r, int t) {
                r.skip = x.p + (long long)t * x.ts;
                r.skip_bs = x.bs;
                r.yout = out.raw ? out.raw + (long long)t * D : nullptr;
                r.yout_elu = out.elu ? out.elu + (long long)t * D : nullptr;
                r.y_bs = (long long)T * D;
            };
            {   // role 0: layer-0 cell step, t = s
                LstmRole& r = q.role[0];
                r.active = s < T;
                r.kind = 0;
                r.a = s > 0 ? ws.hseq0 + (long long)(s - 1) * BD : nullptr;
                r.wpk = h->blob + lp.hh_off[0];
                r.gin = ws.gin + (long long)s * B4D;
                r.hnext = ws.hseq0 + (long long)s * BD;
                r.c = ws.c;
                r.first = s == 0;
                if (L == 1 && r.active) fill_last(r, s);
            }
            if (L == 2) {
                const int t1 = s - 1, t2 = s - 2;
                LstmRole& pr = q.role[1];   // layer-1 input projection, t = s-1
                pr.active = t1 >= 0 && t1 < T;
                pr.kind = 1;
                pr.a = ws.hseq0 + (long long)std::max(t1, 0) * BD;
                pr.wpk = h->blob + lp.ihpk_off[1];
                pr.bias = h->blob + lp.ih[1].b_off;
                pr.gout = ws.gin1 + (long long)std::max(t1, 0) * B4D;
                LstmRole& r = q.role[2];    // layer-1 cell step, t = s-2
                r.active = t2 >= 0 && t2 < T;
                r.kind = 0;
                r.a = t2 > 0 ? ws.hseq1 + (long long)(t2 - 1) * BD : nullptr;
                r.wpk = h->blob + lp.hh_off[1];
                r.gin = ws.gin1 + (long long)std::max(t2, 0) * B4D;
                r.hnext = ws.hseq1 + (long long)std::max(t2, 0) * BD;
                r.c = ws.c + (long long)B * D;
                r.first = t2 == 0;
                if (r.active) fill_last(r, t2);
            }
            const dim3 grid(D / 4, cdiv(B, 32), nroles), block(256);
            switch (D / 32) {
                case 2: hipLaunchKernelGGL(lstm_step_kernel<2>, grid, block, 0, st, q); break;
                case 4: hipLaunchKernelGGL(lstm_step_kernel<4>, grid, block, 0, st, q); break;
                case 8: hipLaunchKernelGGL(lstm_step_kernel<8>, grid, block, 0, st, q); break;
                default: hipLaunchKernelGGL(lstm_step_kernel<16>, grid, block, 0, st, q); break;
            }
        }
    }
    HIPCHK(h, hipGetLastError());
    const unsigned* am = amax_plus(h, st, x, 1.0f, B);
    y->raw = Act{out.raw, (long long)T * D, D, T, D, am, B};
    y->elu = Act{out.elu, (long long)T * D, D, T, D, am, B};
    return AC_OK;
}


int rvq_encode_fwd(ac_handle* h, hipStream_t st, const float* feats, int F, int K, long long* toks) {
    RvqEncParams p{};
    p.x = feats;
    p.epk = h->blob + h->cb_packed;
    p.e = h->blob + h->cb_plain;
    p.ee = h->blob + h->cb_ee;
    p.toks = toks;
    p.F = F;
    p.H = h->cfg.hidden_size;
    p.C = h->cfg.codebook_size;
    p.K = K;
    p.xs = p.H;
    p.tK = K;
    p.tk0 = 0;
    const int HV = p.H / 16;
    // frames per wave: 48 once there are enough frames to fill every SIMD (1024) with one wave
    const int MS = (HV <= 8 && F >= 1024 * 32) ? 3 : 1;
    const dim3 grid(cdiv(F, 16 * MS)), block(64);
    const bool exact_env = h->dev.rvq_exact != 0;   // developer A/B switch
    if (h->cb16 && (HV == 8 || (HV == 32 && K == 1)) && !exact_env) {   // split16 products on the fp16 matrix pipe (rvq16.h)
        RvqEnc16Params q{};
        q.base = p;
        q.epk16 = reinterpret_cast<const _Float16*>(h->blob + h->cb16);
        q.einv = h->blob + h->cb16_inv;
        ProfScope ps(h, st, "rvq_encode16_kernel", 2.0 * F * (double)p.C * p.H * K,
                     (double)F * p.H * 4 + (double)F * K * 8 + (double)K * p.C * p.H * 4);
        if (HV == 32) hipLaunchKernelGGL((rvq_encode16_kernel<32, 1, false, true>), dim3(cdiv(F, 16)), block, 0, st, q);      // WavTokenizer: 4096 x 512, one stage
        else if (MS == 3) hipLaunchKernelGGL((rvq_encode16_kernel<8, 3, false>), grid, block, 0, st, q);
        else if (F <= 4096 && p.C % 128 == 0) hipLaunchKernelGGL((rvq_encode16_kernel<8, 1, false, false, 4>), grid, dim3(256), 0, st, q);   // few frame groups: four waves share each (rvq16.h WS)
        else hipLaunchKernelGGL((rvq_encode16_kernel<8, 1, false>), grid, block, 0, st, q);
        HIPCHK(h, hipGetLastError());
        return AC_OK;
    }
    ProfScope ps(h, st, "rvq_encode_kernel", 2.0 * F * (double)p.C * p.H * K,
                 (double)F * p.H * 4 + (double)F * K * 8 + (double)K * p.C * p.H * 4);
#define RVQ_CASE(HV_, MS_) hipLaunchKernelGGL((rvq_encode_kernel<HV_, MS_, false>), grid, block, 0, st, p)
    switch (HV * 10 + MS) {
        case 11: RVQ_CASE(1, 1); break;
        case 13: RVQ_CASE(1, 3); break;
        case 21: RVQ_CASE(2, 1); break;
        case 23: RVQ_CASE(2, 3); break;
        case 41: RVQ_CASE(4, 1); break;
        case 43: RVQ_CASE(4, 3); break;
        case 81: RVQ_CASE(8, 1); break;
        case 83: RVQ_CASE(8, 3); break;
        case 161: RVQ_CASE(16, 1); break;
        case 321: RVQ_CASE(32, 1); break;
        default: return fail(h, AC_EINVAL, "hidden_size %d unsupported by the RVQ kernel (need 16*{1,2,4,8,16,32})", p.H);
    }
#undef RVQ_CASE
    HIPCHK(h, hipGetLastError());
    return AC_OK;
}


int rvq_decode_fwd(ac_handle* h, hipStream_t st, const long long* toks, int F, int K, float* out) {
    RvqDecParams p{};
    p.toks = toks;
    p.e = h->blob + h->cb_plain;
    p.out = out;
    p.F = F;
    p.H = h->cfg.hidden_size;
    p.C = h->cfg.codebook_size;
    p.K = K;
    p.tK = K;
    p.tk0 = 0;
    p.os = p.H;
    const long long n = (long long)F * (p.H / 4);
    p.bad = h->sticky_dev ? h->sticky_dev + ST_BAD_TOKEN : nullptr;
    ProfScope ps(h, st, "rvq_decode_kernel", (double)F * p.H * K, (double)F * K * 8 + (double)F * p.H * 4 * (K + 1));
    hipLaunchKernelGGL(rvq_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    HIPCHK(h, hipGetLastError());
    return AC_OK;
}


Workspace plan_ws(const ac_handle* h, int B, int T_in /*samples, encoder*/, int N_frames /*decoder*/, bool enc) {
    const ac_config& c = h->cfg;
    Workspace w;
    size_t mx = 0;
    int N;
    if (enc) {
        long long L = T_in;
        int ch = c.num_filters;
        mx = std::max(mx, (size_t)L * ch);
        for (int r = c.num_ratios - 1; r >= 0; --r) {
            L = (L + c.upsampling_ratios[r] - 1) / c.upsampling_ratios[r];
            ch *= 2;
            mx = std::max(mx, (size_t)L * ch);
        }
        N = (int)L;
    } else {
        N = N_frames;
        long long L = N;
        int ch = h->D;
        mx = std::max(mx, (size_t)L * ch);
        for (int r = 0; r < c.num_ratios; ++r) {
            L *= c.upsampling_ratios[r];
            ch /= 2;
            mx = std::max(mx, (size_t)L * ch);
        }
    }
    mx = std::max(mx, (size_t)N * std::max(h->D, c.hidden_size));
    w.act_floats = align_up(mx * B, 64);
    w.hseq = align_up((size_t)N * ((B + 31) / 32 * 32) * h->D * 3 / 2, 64);   // clips padded to the 32-clip workgroup tile; x1.5: room of the removed three-plane layout kept (workspace sizes unchanged); lstm_persist16.h's fp16 plane blocks need x1
    w.gin = std::max(align_up((size_t)N * B * 4 * h->D, 64), w.hseq);          // gin1 doubles as layer 0's local h copy on the persistent path
    w.c = align_up((size_t)2 * B * h->D, 64);
    w.total_bytes = (NACT * w.act_floats + 2 * w.gin + 2 * w.hseq + w.c) * sizeof(float) + 256;
    add_pool(w, B, 0);
    return w;
}


int carve(ac_handle* h, const Workspace& w, void* ws, size_t ws_bytes, WsPtrs* o) {
    if (!ws) return fail(h, AC_EINVAL, "workspace pointer is null");
    if (ws_bytes < w.total_bytes) return fail(h, AC_ENOMEM, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_bytes);
    char* base = reinterpret_cast<char*>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
    pool_bind(h, base, w.pool_B, w.pool_rows);
    float* p = reinterpret_cast<float*>(base + align_up(pool_bytes(w.pool_B, w.pool_rows), 256));
    for (int i = 0; i < NACT; ++i) { o->act[i] = p; o->used[i] = false; p += w.act_floats; }
    o->lstm.gin = p; p += w.gin;
    o->lstm.gin1 = p; p += w.gin;
    o->lstm.hseq0 = p; p += w.hseq;
    o->lstm.hseq1 = p; p += w.hseq;
    o->lstm.c = p;
    return AC_OK;
}


// Failures a kernel can only detect on the device surface here, at the NEXT entry point of the handle (no entry point
// synchronises): the failed call's outputs were set to NaN by the device, never left unwritten.
int check_ready(ac_handle* h) {
    if (!h) return AC_EINVAL;
    if (!h->finalized) return fail(h, AC_ESTATE, "ac_finalize has not been called");
    if (h->sticky) {
        // EVERY pending class is reported and cleared by this one call (round-3 advisor finding: with one class per call a bad-token
        // count that was pending beside an LSTM failure surfaced in a later, unrelated call -- what strict mode exists to prevent)
        volatile unsigned* s = h->sticky;
        const unsigned a = s[ST_LSTM_TIMEOUT], b = s[ST_LSTM_PLACEMENT], t = s[ST_BAD_TOKEN];
        if (a || b || t) {
            s[ST_LSTM_TIMEOUT] = 0;
            s[ST_LSTM_PLACEMENT] = 0;
            s[ST_BAD_TOKEN] = 0;
            char lstm[400] = "", tok[200] = "";
            if (a || b) {
                h->lstm_step_only = true;   // self-heal: the per-step kernels need no co-residency
                std::snprintf(lstm, sizeof lstm,
                              "an EARLIER call's persistent LSTM launch failed (%u bounded waits expired, %u launches without 32 workgroups on "
                              "every XCD -- is the GPU shared?): that call's outputs were set to NaN; the handle now uses the per-step LSTM "
                              "kernels, repeat the call", a, b);
            }
            if (t) std::snprintf(tok, sizeof tok, "an EARLIER ac_decode / ac_dequantize call got %u token ids outside [0, codebook_size): those frames were set to NaN", t);
            return fail(h, (a || b) ? AC_EHIP : AC_EINVAL, "%s%s%s", lstm, (a || b) && t ? "; ALSO: " : "", tok);
        }
    }
    return AC_OK;
}


// The kernels address one clip's activation with 32-bit byte offsets (buffer descriptors): the widest per-clip tensor
// (64 floats per sample; DAC: up to 128) must stay below 2 GB -- 7.3 M samples (5 min at 24 kHz) per clip and call,
// DAC 3.6 M (83 s at 44.1 kHz).
int check_len(ac_handle* h, long long samples) {
    const long long lim = 0x70000000LL / (h->arch == ARCH_DAC ? 512 : 256) - 1;
    if (samples > lim) return fail(h, AC_EINVAL, "clip of %lld samples is too long for one call (limit %lld): split it", samples, lim);
    return AC_OK;
}


// encoder: sig -> feats [B][N][H] written to `feats`.
// Flavours: a tensor is written raw where a shortcut / LSTM / caller reads it, ELU'd where the next
// conv reads it (all SEANet convs but the first are preceded by nn.ELU), both where both happen.
// While the test hook is armed every module output is also written raw so it can be captured.
int encoder_fwd(ac_handle* h, hipStream_t st, const float* sig, const float* rel_len, int B, int T, float* feats, WsPtrs& ws) {
    const ac_config& c = h->cfg;
    const bool dbg = h->dbg != nullptr;
    Act xin{sig, (long long)T, 1, T, 1};
    Act2 x, y;
    int rc;
    // a 32-channel ResBlock activates its raw input itself (rb_fused.h): no ELU'd flavour needed in HBM
    auto rb_self_elu = [&](int C) {
        if (h->noncausal && h->gemm_fp32) return false;   // non-causal blocks are fused in split-operand arithmetic only
        return (C == 32 || C == 64 || (C == 128 && c.num_ratios > 2 && rb128_ok(h, h->enc_rb[2]))) && c.residual_kernel_size == 3 && c.compress == 2;
    };
    int i0 = 0;
    if (enc_front_ok(h, T)) {   // stem, first residual block and first down-sampler as one kernel (enc_front.h)
        float* d0 = dbg ? ws.take() : nullptr;
        float* d1 = dbg ? ws.take() : nullptr;
        rc = enc_front_fwd(h, st, sig, rel_len, B, T, ws.take(), d0, d1, &x);
        if (rc) return rc;
        if (dbg) {   // the module outputs inside the chain, written by the same kernel while the test hook is armed
            capture(h, st, Act{d0, (long long)T * 32, 32, T, 32}, B);
            capture(h, st, Act{d1, (long long)T * 32, 32, T, 32}, B);
            ws.give(d0);
            ws.give(d1);
        }
        capture(h, st, x.raw, B);
        i0 = 1;
    } else {
    if (thin_ok(c, c.kernel_size))
        rc = stem_fwd(h, st, sig, rel_len, B, T, Out{ws.take(), rb_self_elu(c.num_filters) ? nullptr : ws.take()}, &x);
    else
        rc = conv_fwd(h, st, h->enc_stem, xin, c.kernel_size, 1, rel_len, Out{ws.take(), ws.take()},
                      (long long)T * c.num_filters, c.num_filters, B, &x);
    if (rc) return rc;
    capture(h, st, x.raw, B);
    }
    for (int i = i0; i < c.num_ratios; ++i) {
        const int ratio = c.upsampling_ratios[c.num_ratios - 1 - i];
        float* hb = ws.take();
        rc = resblock_fwd(h, st, h->enc_rb[i], x, hb, Out{dbg ? ws.take() : nullptr, ws.take()}, B, &y);
        if (rc) return rc;
        ws.give(hb);
        ws.give(x);
        if (dbg) { capture(h, st, y.raw, B); ws.give(y.raw.p); y.raw.p = nullptr; }
        x = y;
        const int M = cdiv(x.elu.L, ratio);
        const bool last = i == c.num_ratios - 1;    // the last down-sampler feeds the LSTM: raw only
        rc = conv_fwd(h, st, h->enc_down[i], x.elu, 2 * ratio, ratio, nullptr,
                      Out{ws.take(), (last || rb_self_elu(h->enc_down[i].N)) ? nullptr : ws.take()},
                      (long long)M * h->enc_down[i].N, h->enc_down[i].N, B, &y);
        if (rc) return rc;
        ws.give(x);
        x = y;
        capture(h, st, x.raw, B);
    }
    rc = lstm_fwd(h, st, h->enc_lstm, x.raw, ws.lstm, Out{dbg ? ws.take() : nullptr, ws.take()}, B, &y);
    if (rc) return rc;
    ws.give(x);
    if (dbg) { capture(h, st, y.raw, B); ws.give(y.raw.p); y.raw.p = nullptr; }
    x = y;
    rc = conv_fwd(h, st, h->enc_final, x.elu, c.last_kernel_size, 1, nullptr, Out{feats, nullptr},
                  (long long)x.elu.L * c.hidden_size, c.hidden_size, B, nullptr);
    ws.give(x);
    return rc;
}


int decoder_fwd(ac_handle* h, hipStream_t st, const long long* toks, int B, int N, int K, float* sig, WsPtrs& ws) {
    const ac_config& c = h->cfg;
    const bool dbg = h->dbg != nullptr;
    float* zb = ws.take();
    int rc = rvq_decode_fwd(h, st, toks, B * N, K, zb);
    if (rc) return rc;
    Act z{zb, (long long)N * c.hidden_size, c.hidden_size, N, c.hidden_size};
    Act2 x, y;
    rc = conv_fwd(h, st, h->dec_first, z, c.kernel_size, 1, nullptr, Out{ws.take(), nullptr}, (long long)N * h->D, h->D, B, &x);
    if (rc) return rc;
    ws.give(zb);
    capture(h, st, x.raw, B);
    rc = lstm_fwd(h, st, h->dec_lstm, x.raw, ws.lstm, Out{dbg ? ws.take() : nullptr, ws.take()}, B, &y);
    if (rc) return rc;
    ws.give(x);
    if (dbg) { capture(h, st, y.raw, B); ws.give(y.raw.p); y.raw.p = nullptr; }
    x = y;
    for (int i = 0; i < c.num_ratios; ++i) {
        if (i == c.num_ratios - 1 && dec_tail_ok(h, x.elu)) {   // last up-sampler, last residual block and the head as one kernel (dec_tail.h)
            const long long T2 = 2LL * x.elu.L;
            float* d0 = dbg ? ws.take() : nullptr;
            float* d1 = dbg ? ws.take() : nullptr;
            rc = dec_tail_fwd(h, st, x.elu, B, sig, d0, d1);
            if (rc) return rc;
            if (dbg) {   // the module outputs inside the chain, written by the same kernel while the test hook is armed
                capture(h, st, Act{d0, T2 * 32, 32, (int)T2, 32}, B);
                capture(h, st, Act{d1, T2 * 32, 32, (int)T2, 32}, B);
                ws.give(d0);
                ws.give(d1);
            }
            ws.give(x);
            return AC_OK;
        }
        const int cup = h->dec_up[i].N / c.upsampling_ratios[i];
        const bool self_elu = (cup == 32 || cup == 64 || (cup == 128 && rb128_ok(h, h->dec_rb[i]))) && c.residual_kernel_size == 3 &&
                              c.compress == 2;   // rb_fused.h / rb_fused6*.h activate raw rows themselves
        rc = convtr_fwd(h, st, h->dec_up[i], x.elu, c.upsampling_ratios[i], Out{ws.take(), self_elu ? nullptr : ws.take()}, B, &y);
        if (rc) return rc;
        ws.give(x);
        x = y;
        capture(h, st, x.raw, B);
        float* hb = ws.take();
        rc = resblock_fwd(h, st, h->dec_rb[i], x, hb, Out{dbg ? ws.take() : nullptr, ws.take()}, B, &y);
        if (rc) return rc;
        ws.give(hb);
        ws.give(x);
        if (dbg) { capture(h, st, y.raw, B); ws.give(y.raw.p); y.raw.p = nullptr; }
        x = y;
    }
    if (thin_ok(c, c.last_kernel_size) && x.elu.ts == c.num_filters)
        rc = head_fwd(h, st, x.elu, B, sig);
    else
        rc = conv_fwd(h, st, h->dec_head, x.elu, c.last_kernel_size, 1, nullptr, Out{sig, nullptr}, (long long)x.elu.L, 1, B, nullptr);
    ws.give(x);
    return rc;
}

// y[rows][N] = epi(x[rows][cin-slice] * W^T): a 1-tap GEMM over the merged token matrix.  `x_pitch` is the row
// pitch of x, `kofs`/`Ktot` select a column block of a wider packed weight.  Rows are chunked so that one
// launch's operands stay below the 2 GB range of a buffer descriptor.
int mimi_linear(ac_handle* h, hipStream_t st, const PackedGemm& g, const float* x, long long rows, int cin, int x_pitch, int kofs,
                float* y, int y_pitch, const Epi& epi) {
    const long long widest = std::max<long long>(std::max(x_pitch, y_pitch), g.N);
    long long chunk = ((1LL << 31) / 4 - 4096) / widest / 128 * 128;
    if (chunk < 128) return fail(h, AC_EINVAL, "linear layer too wide for the tap-GEMM (%lld)", widest);
    for (long long r0 = 0; r0 < rows; r0 += chunk) {
        const long long n = std::min(chunk, rows - r0);
        TapGemmParams p{};
        p.nseg = 1;
        Act xa{x + r0 * x_pitch, 0, x_pitch, (int)n, cin};
        const bool one = r0 == 0 && n == rows;     // row words are chained between launches that cover the whole matrix
        if (one && epi.rowmax_in) { xa.amax = epi.rowmax_in; xa.amax_n = -(int)n; }
        p.seg[0] = make_seg(xa, 1, 1, PAD_ZERO, 0, kofs, nullptr);
        if (one && epi.rowmax_out) p.amax_out_rows = reinterpret_cast<unsigned*>(1);   // request; run_tap allocates
        p.amax_rows = 1;                           // split16.h row mode: every row scales by its own amax
        p.w = h->blob + g.w_off;
        p.bias = g.has_bias ? h->blob + g.b_off : nullptr;
        p.y = y + r0 * y_pitch;
        p.y_bs = 0;
        p.y_rs = y_pitch;
        p.B = 1;
        p.M = (int)n;
        p.N = g.N;
        p.Ktot = g.Ktot;
        p.scale = epi.scale;
        p.res = epi.res ? epi.res + r0 * epi.res_rs : nullptr;
        p.res_bs = 0;
        p.res_rs = epi.res_rs;
        p.gelu = epi.gelu;
        if (int rc = run_tap(h, st, p)) return rc;
        if (epi.rowmax_out) *epi.rowmax_out = one ? p.amax_out_rows : nullptr;
    }
    return AC_OK;
}

int upload_blob(ac_handle* h, Packer& pk, int device) {
    HIPCHK(h, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(h, AC_ENODEV, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    h->num_cus = prop.multiProcessorCount;
    const char* lm = std::getenv("AC_LSTM");
    h->lstm_step_only = lm && std::strcmp(lm, "step") == 0;
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->lp_ctl), LP_CTL_WORDS * sizeof(unsigned)));
    HIPCHK(h, hipMemset(h->lp_ctl, 0, LP_CTL_WORDS * sizeof(unsigned)));
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->sticky), ST_WORDS * sizeof(unsigned), hipHostMallocMapped));
    std::memset(h->sticky, 0, ST_WORDS * sizeof(unsigned));
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->sticky_dev), h->sticky, 0));
    if (h->num_cus == 256 && !h->lstm_step_only) {
        // the persistent LSTM assigns roles from XCC_ID: use it only where a 256-workgroup cooperative launch really
        // lands 32 workgroups on each of 8 XCDs (otherwise: the per-step kernel)
        unsigned* hist = h->lp_ctl;
        void* args[] = {&hist};
        unsigned got[16] = {0};
        bool ok = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(lstm_persist_probe_kernel), dim3(256), dim3(256), args, 0, nullptr) == hipSuccess &&
                  hipDeviceSynchronize() == hipSuccess && hipMemcpy(got, hist, sizeof got, hipMemcpyDeviceToHost) == hipSuccess;
        for (int i = 0; ok && i < 16; ++i) ok = got[i] == (i < 8 ? 32u : 0u);
        if (!ok) { (void)hipGetLastError(); h->lstm_step_only = true; }
        HIPCHK(h, hipMemset(h->lp_ctl, 0, LP_CTL_WORDS * sizeof(unsigned)));
    }
    if (h->arch == ARCH_MIMI) {
        h->own_pool_rows = (size_t)h->mcfg.codebook_size;
        HIPCHK(h, hipMalloc(&h->own_pool, pool_bytes(1, h->own_pool_rows)));
    }
    h->blob_floats = pk.blob.size();
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->blob), h->blob_floats * sizeof(float)));
    HIPCHK(h, hipMemcpy(h->blob, pk.blob.data(), h->blob_floats * sizeof(float), hipMemcpyHostToDevice));
    h->host.clear();
    h->finalized = true;
    return AC_OK;
}

// ---- wrappers for the per-codec translation units
// Mimi's last decoder block with the final Conv1d(64, 1, k) folded in (rb_fused6.h HEAD); false: the shape / mode wants the two kernels
bool rb64_identity_head_fwd(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, const PackedGemm& head, int head_k, float* sig, int B, int* rc) {
    if (!rb.has6 || h->gemm_fp32 || h->dbg || h->dev.mimi_tail == 0 || head_k < 1 || head_k > 8 || head.N != 1 || head.Ktot != head_k * 64 || !head.has_bias) return false;
    *rc = launch_rb_fused6<64, false>(h, st, rb, x, Out{}, B, PAD_ZERO, nullptr, &head, head_k, sig);
    return true;
}
int rb64_identity_fwd(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, Out out, int B, const unsigned** amax_out) {
    return rb.has6 && !h->gemm_fp32 ? launch_rb_fused6<64, false>(h, st, rb, x, out, B, PAD_ZERO, amax_out) : launch_rb_fused<64, 64, 2, false>(h, st, rb, x, out, B, PAD_ZERO);
}
int rb128_identity_fwd(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, Out out, int B, const unsigned** amax_out) {
    return launch_rb128_fused6<false>(h, st, rb, x, out, B, PAD_ZERO, amax_out);
}
int rvq_encode_cdist_launch(ac_handle* h, hipStream_t st, const RvqEncParams& p, unsigned blocks, const _Float16* epk16, const float* einv) {
    const dim3 grid(blocks), block(64);
    if (epk16 && p.H == 256 && !h->dev.rvq_exact) {      // Mimi: 2048 x 256 tables in split16 arithmetic (rvq16.h)
        RvqEnc16Params q{};
        q.base = p;
        q.epk16 = epk16;
        q.einv = einv;
        hipLaunchKernelGGL((rvq_encode16_kernel<16, 1, true>), grid, block, 0, st, q);
        return AC_OK;
    }
    switch (p.H / 16) {
        case 1: hipLaunchKernelGGL((rvq_encode_kernel<1, 1, true>), grid, block, 0, st, p); break;
        case 2: hipLaunchKernelGGL((rvq_encode_kernel<2, 1, true>), grid, block, 0, st, p); break;
        case 4: hipLaunchKernelGGL((rvq_encode_kernel<4, 1, true>), grid, block, 0, st, p); break;
        case 8: hipLaunchKernelGGL((rvq_encode_kernel<8, 1, true>), grid, block, 0, st, p); break;
        case 16: hipLaunchKernelGGL((rvq_encode_kernel<16, 1, true>), grid, block, 0, st, p); break;
        default: return fail(h, AC_EINVAL, "codebook_dim %d unsupported by the RVQ kernel (need 16*{1,2,4,8,16})", p.H);
    }
    return AC_OK;
}
void rvq_decode_launch(hipStream_t st, const RvqDecParams& p, unsigned blocks) {
    hipLaunchKernelGGL(rvq_decode_kernel, dim3(blocks), dim3(256), 0, st, p);
}
int resample_launch(const float* x, int B, int L, const float* kern, int n, int o, int taps, int width, float* y, int L_out, hipStream_t st) {
    ResampleParams p{x, kern, y, B, L, L_out, n, o, taps, width};
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((L_out + 255) / 256), B), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? AC_OK : AC_EHIP;
}
void amax_fill_launch(hipStream_t st, unsigned* slot, unsigned bits, int B) {
    hipLaunchKernelGGL(amax_fill_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, slot, bits, B);
}

}  // namespace acimpl
