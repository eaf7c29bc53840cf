// C ABI of the MI355X EnCodec path (include/audiocodecs_amd.h): model plan, weight packing,
// workspace layout and the launch sequences of encode / decode.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/audiocodecs_amd.h"
#include "lstm.h"
#include "lstm_persist.h"
#include "lstm_persist6.h"
#include "lstm_persist16.h"
#include "rvq.h"
#include "rvq16.h"
#include "tap_gemm.h"
#include "tap_gemm4.h"
#include "tap_gemm6.h"
#include "thin.h"
#include "rb_fused.h"
#include "rb_fused6.h"
#include "thin_conv6.h"
#include "rb_fused6_128.h"
#include "enc_front.h"
#include "dec_tail.h"
#include "mimi.h"
#include "dac.h"
#include "wavtok.h"

using namespace ac;

namespace {

struct ConvSpec {
    std::string prefix;
    int transposed, cin, cout, k, s;
};

struct PackedGemm {      // one tap_gemm launch worth of weights
    size_t w_off = 0, b_off = 0;  // float offsets into the device blob
    int N = 0, Ktot = 0;
    bool has_bias = true;
};

struct ResBlockPlan {
    int C;
    PackedGemm c3;     // k3 conv C -> C/2
    PackedGemm fused;  // [ELU(h) | x] * [W1; Ws] + (b1 + bs)
    size_t w3f_off = 0, wff_off = 0;   // rb_fused6.h fragment images of the two matrices (float offsets into the blob)
    size_t winv3_off = 0, winvf_off = 0;   // split16.h: per-row 2^-s of the two images
    float hb0 = 0.f, hb1 = 0.f;            // split16.h: |hidden| <= hb0 + hb1 * amax(x)
    bool has6 = false;
};

struct LstmPlan {
    int D, layers;
    std::vector<PackedGemm> ih;     // [4D][D] + (b_ih + b_hh)
    std::vector<size_t> hh_off;     // W_hh per layer, MFMA B-fragment order
    std::vector<size_t> ihpk_off;   // W_ih per layer, same order (used by the in-step projection of layers >= 1)
    size_t persist_off = 0;         // register images of W_hh0, W_ih1, W_hh1 for lstm_persist_kernel (D = 512, 2 layers)
    size_t persist6_off = 0;        // the same as three bf16 planes for lstm_persist6_kernel (float offset into the blob)
    size_t persist16_inv = 0;       // split16.h: persist6_off holds two fp16 planes of the scaled rows; [2 layers][4D] 2^-s
    bool has_persist = false;
};

struct MimiTfLayer {
    PackedGemm qkv, o, fc1, fc2;                       // [3A][H], [H][A], [I][H], [H][I]; no biases
    size_t ln1_w = 0, ln1_b = 0, ln2_w = 0, ln2_b = 0, sc_a = 0, sc_m = 0;
};

struct MimiPlan {
    PackedGemm enc_stem, enc_final, down, dec_first, dec_head;
    std::vector<ResBlockPlan> enc_rb, dec_rb;          // fused = the k1 conv alone (identity shortcut)
    std::vector<PackedGemm> enc_down, dec_up;
    std::vector<MimiTfLayer> enc_tf, dec_tf;
    size_t up_w = 0;                                   // depthwise transposed conv [H][2*stride]
    PackedGemm in_proj;                                // [2*Dq][H]: semantic rows, then acoustic rows
    PackedGemm out_proj;                               // [H][2*Dq]: semantic | acoustic columns
    size_t cb_plain = 0, cb_packed = 0, cb_ee = 0;     // [Q][C][Dq] in wrapper order (semantic first)
    size_t rope_cos = 0, rope_sin = 0;                 // [rope_T][head_dim]
    int rope_T = 0;
    int D = 0;                                         // SEANet width at the bottleneck
};

struct DacResUnitPlan {
    PackedGemm c7, c1;                                 // dilated k7 conv, k1 conv (both C -> C)
    size_t a1 = 0, a1i = 0, a2 = 0, a2i = 0;           // Snake alpha / (alpha + 1e-9)^-1 of snake1, snake2
    int dil = 1;
};

struct DacBlockPlan {
    int C = 0, stride = 1;                             // residual-unit width; stride of the block's (transposed) conv
    std::vector<DacResUnitPlan> ru;
    size_t a = 0, ai = 0;                              // the block's own Snake (before the strided / transposed conv)
    PackedGemm conv;
};

struct DacPlan {
    PackedGemm enc_stem, enc_final, dec_first, dec_head, in_proj0;
    std::vector<DacBlockPlan> enc, dec;
    size_t enc_a = 0, enc_ai = 0, dec_a = 0, dec_ai = 0;
    size_t win = 0, bin = 0, wout = 0, bout = 0, cb = 0, cbn = 0, c2 = 0, proj = 0;
    int H = 0;                                         // latent width
};

struct WtResnetPlan {
    size_t n1w = 0, n1b = 0, n2w = 0, n2b = 0;
    PackedGemm c1, c2;
};
struct WtCnxPlan {
    size_t dww = 0, dwb = 0, sc = 0, sh = 0, gamma = 0;
    PackedGemm p1, p2;
};
struct WavtokPlan {
    PackedGemm embed, qkv, proj, head, istft;
    WtResnetPlan rn[4];                       // pos_net.0, .1, .3, .4
    size_t an_w = 0, an_b = 0;                // pos_net.2.norm
    size_t g5w = 0, g5b = 0;                  // pos_net.5
    size_t nsc = 0, nsh = 0;                  // backbone.norm rows `bandwidth_id`
    size_t flw = 0, flb = 0;                  // final_layer_norm
    size_t w2 = 0;                            // squared window [nfft]
    std::vector<WtCnxPlan> cnx;
    int bins = 0, npad = 0, hop_pad = 0, taps = 0;
};

struct ProfRec {
    int name_id;
    int count;
    hipEvent_t e0, e1;
    double flops, bytes;
};

}  // namespace

enum { ARCH_ENCODEC = 0, ARCH_MIMI = 1, ARCH_DAC = 2, ARCH_WAVTOK = 3 };

struct ac_handle {
    int arch = ARCH_ENCODEC;
    ac_config cfg{};
    ac_mimi_config mcfg{};
    MimiPlan mimi;
    ac_dac_config dcfg{};
    DacPlan dac;
    ac_wavtok_config wcfg{};
    WavtokPlan wt;
    std::string err;
    std::map<std::string, std::vector<float>> host;
    bool finalized = false;
    float* blob = nullptr;
    size_t blob_floats = 0;
    int hop = 1, D = 0;
    // encoder plan
    PackedGemm enc_stem, enc_final;
    std::vector<ResBlockPlan> enc_rb;
    std::vector<PackedGemm> enc_down;
    LstmPlan enc_lstm, dec_lstm;
    // decoder plan
    PackedGemm dec_first, dec_head;
    std::vector<PackedGemm> dec_up;
    std::vector<ResBlockPlan> dec_rb;
    // codebooks
    size_t cb_plain = 0, cb_packed = 0, cb_ee = 0;
    size_t cb16 = 0, cb16_inv = 0;   // rvq16.h: split16 images of the codebooks + their 2^-s (0: not packed -- other arithmetic or shape)
    // bounds of the fused thin-channel chains (enc_front.h): |stem out| <= sb0 + sb1 amax(sig); |block out| <= fb0 + fb1h H + fb1x X
    struct ChainBounds { float sb0 = 0.f, sb1 = 0.f, fb0 = 0.f, fb1h = 0.f, fb1x = 0.f; bool ok = false; } enc_front, dec_tail;   // dec_tail: sb0 / sb1 are the transposed conv's
    bool fuse_chains = true;        // AC_FUSE=0 at ac_finalize: the layers of the fused chains as separate kernels (A/B runs, cross-check tests)
    // test hook: copy every layer output (standard [B][L][C] layout) into a caller buffer
    float* dbg = nullptr;
    size_t dbg_cap = 0, dbg_used = 0;
    // kernels that already got their > 64 KB dynamic-LDS opt-in on this handle's device
    std::vector<const void*> lds_opted;
    // split-operand weights (tap_gemm6.h): float offset of a packed fp32 matrix -> float offset of its bf16 planes
    std::map<size_t, size_t> w6_of;
    std::map<size_t, size_t> t6_of;   // thin_conv6.h fragment images of the [64][128] layers, keyed like w6_of
    std::map<size_t, size_t> t6inv_of;   // split16.h: their per-row 2^-s
    bool noncausal = false;                // WavTokenizer's SEANet encoder: centred padding (right = total/2, left = total - right)
    bool has_enc = true, has_dec = true;   // a half the caller's mode never runs may be left out (encodec.py:67-71)
    bool gemm_fp32 = false;         // AC_PRECISION_FP32_EXACT (or AC_GEMM=fp32): exact-product kernels only
    bool gemm_bf16 = false;         // AC_PRECISION_BF16 (or AC_GEMM=bf16): opt-in, operands rounded to bf16 in the tap-GEMMs
    bool split16 = true;            // fp32-fidelity arithmetic of the matrix kernels: two fp16 planes, 3 products (split16.h);
                                    // false (AC_PRECISION_FP32_BF16X3 / AC_SPLIT=bf16x3): three bf16 planes, 6 products
    std::map<size_t, size_t> winv_of;   // split16 images: float offset of a packed fp32 matrix -> offset of its per-row 2^-s
    // amax slots (split16.h): [slot][amax_B] words, handed out in launch order, cleared at the start of every pass
    unsigned* amax_buf = nullptr;
    int amax_B = 0, amax_next = 0;
    // row mode (linear layers over merged row matrices): a ring of per-row words
    unsigned* row_buf = nullptr;
    size_t row_cap = 0, row_next = 0;
    // the only pool the handle owns: a few KB allocated at ac_finalize for ac_embs_projected (Mimi), the one launching entry
    // point without a workspace argument
    void* own_pool = nullptr;
    size_t own_pool_rows = 0;
    int precision = -1;             // ac_set_precision; -1: take AC_GEMM from the environment
    // persistent LSTM (lstm_persist.h): control words, device shape, opt-out (AC_LSTM=step)
    unsigned* lp_ctl = nullptr;
    int num_cus = 0;
    bool lstm_step_only = false;
    // sticky status words (lstm_persist.h ST_*): host-pinned, device-mapped -- read on the host without synchronising
    unsigned long long* clk_dev = nullptr;   // ac_debug_clock: shader / real-time tick sums of the tap_gemm6 workgroups
    unsigned* sticky = nullptr;       // host view
    unsigned* sticky_dev = nullptr;   // device view of the same words
    // profiling
    bool prof = false;
    bool prof_detail = false;   // AC_PROF_DETAIL=1: one record per tap-GEMM shape
    std::vector<ProfRec> recs;
    std::vector<std::string> prof_names;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
};

namespace {

int fail(ac_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return fail(h, AC_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// More than 64 KB of dynamic LDS must be opted into per kernel (and per device): once per handle.
int ensure_lds(ac_handle* h, const void* func, size_t bytes) {
    if (bytes <= 64 * 1024) return AC_OK;
    for (const void* f : h->lds_opted)
        if (f == func) return AC_OK;
    HIPCHK(h, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    h->lds_opted.push_back(func);
    return AC_OK;
}
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// architecture enumeration (HF module order)
// ---------------------------------------------------------------------------------------------
struct Arch {
    std::vector<ConvSpec> enc_rb3, enc_rb1, enc_rbs, enc_down;  // per stage
    ConvSpec enc_stem, enc_final, dec_first, dec_head;
    std::vector<ConvSpec> dec_up, dec_rb3, dec_rb1, dec_rbs;
    std::string enc_lstm, dec_lstm;
    int D;
};

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

// ---------------------------------------------------------------------------------------------
// weight access + packing
// ---------------------------------------------------------------------------------------------
struct Packer {
    ac_handle* h;
    std::vector<float> blob;
    int rc = AC_OK;

    const std::vector<float>* get(const std::string& name, size_t n) {
        auto it = h->host.find(name);
        if (it == h->host.end()) {
            rc = fail(h, AC_ESTATE, "missing tensor '%s'", name.c_str());
            return nullptr;
        }
        if (it->second.size() != n) {
            rc = fail(h, AC_EINVAL, "tensor '%s' has %zu elements, expected %zu", name.c_str(), it->second.size(), n);
            return nullptr;
        }
        return &it->second;
    }
    // conv weight in HF layout ([cout][cin][k], or [cin][cout][k] when transposed), folding
    // weight-norm when only (g, v) were given: w = v * (g / ||v||_2), norm over dims (1,2).
    bool weight(const ConvSpec& s, std::vector<float>& w) {
        const size_t n = (size_t)s.cin * s.cout * s.k;
        auto it = h->host.find(s.prefix + ".weight");
        if (it != h->host.end()) {
            if (it->second.size() != n) {
                rc = fail(h, AC_EINVAL, "tensor '%s.weight' has %zu elements, expected %zu", s.prefix.c_str(), it->second.size(), n);
                return false;
            }
            w = it->second;
            return true;
        }
        const int d0 = s.transposed ? s.cin : s.cout;
        const std::vector<float>* g = get(s.prefix + ".parametrizations.weight.original0", d0);
        if (!g) return false;
        const std::vector<float>* v = get(s.prefix + ".parametrizations.weight.original1", n);
        if (!v) return false;
        w.resize(n);
        const size_t inner = n / d0;
        for (int i = 0; i < d0; ++i) {
            double ss = 0.0;
            for (size_t j = 0; j < inner; ++j) ss += (double)(*v)[i * inner + j] * (*v)[i * inner + j];
            const float scale = (*g)[i] / (float)std::sqrt(ss);
            for (size_t j = 0; j < inner; ++j) w[i * inner + j] = (*v)[i * inner + j] * scale;
        }
        return true;
    }
    size_t reserve(size_t n) {
        const size_t off = align_up(blob.size(), 64);
        blob.resize(off + n, 0.f);
        return off;
    }
    // tap_gemm6.h weight operand: exact truncation split of every weight into three bf16 terms, packed in MFMA
    // B-fragment order  [n-tile of 32][k-step of 16][plane][lane 64][8]
    static uint16_t bf16_rn(float v) {
        uint32_t b;
        std::memcpy(&b, &v, 4);
        if ((b & 0x7f800000u) == 0x7f800000u) return (uint16_t)(b >> 16);   // inf / nan
        return (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
    }
    // fp16 round-to-nearest-even (denormals kept, overflow -> inf) and back: split16.h on the host
    static uint16_t f16_rn(float f) {
        uint32_t x;
        std::memcpy(&x, &f, 4);
        const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
        x &= 0x7fffffffu;
        if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
        if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);      // >= 65520 rounds to inf
        if (x <= 0x33000000u) return sign;                           // <= 2^-25: half the smallest denormal ties to even = 0
        const int e = (int)(x >> 23) - 127;
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = e >= -14 ? 13 : 13 + (-14 - e);
        uint32_t q = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) ++q;
        if (e >= -14) return (uint16_t)(sign | (uint16_t)(((uint32_t)(e + 15) << 10) + (q - 0x400u)));
        return (uint16_t)(sign | (uint16_t)q);
    }
    static float f16_f32(uint16_t hbits) {
        const int e = (hbits >> 10) & 31, m = hbits & 0x3ff;
        float v;
        if (e == 0) v = std::ldexp((float)m, -24);
        else if (e == 31) v = m ? NAN : INFINITY;
        else v = std::ldexp((float)(m | 0x400), e - 25);
        return (hbits & 0x8000) ? -v : v;
    }
    // split16.h scale of one weight row: 2^s with |w| 2^s < 2^15; returns s
    static int row_scale(const float* w, size_t n, const std::vector<int>* kmap = nullptr) {
        uint32_t mx = 0;
        for (size_t k = 0; k < (kmap ? kmap->size() : n); ++k) {
            if (kmap && (*kmap)[k] < 0) continue;
            uint32_t b;
            std::memcpy(&b, &w[kmap ? (size_t)(*kmap)[k] : k], 4);
            b &= 0x7fffffffu;
            mx = std::max(mx, b);
        }
        return s16_exponent(mx, 40);
    }
    static void split16h(float v, int s, uint16_t (&o)[3]) {
        const float vs = std::ldexp(v, s);
        o[0] = f16_rn(vs);
        o[1] = f16_rn(vs - f16_f32(o[0]));
        o[2] = 0;
    }
    bool use16() const { return h->split16 && !h->gemm_bf16 && !h->gemm_fp32; }
    // the LSTM stays fp32-faithful in the opt-in bf16 mode: split16 there too (three bf16 planes only in AC_PRECISION_FP32_BF16X3)
    bool lstm16() const { return !h->gemm_fp32 && (h->split16 || h->gemm_bf16); }
    // tap_gemm6 NP = 2 image: [n-tile of 32][k-step of 16][plane 2][lane 64][8 fp16] of the scaled rows + winv[N]
    void pack16(const PackedGemm& g) {
        const size_t n_el = (size_t)g.N * g.Ktot;
        const size_t off = reserve(n_el);
        const size_t ioff = reserve(g.N);
        std::vector<uint16_t> planes(2 * n_el);
        const int ksteps = g.Ktot / 16;
        std::vector<int> sc(g.N);
        for (int n = 0; n < g.N; ++n) {
            sc[n] = row_scale(&blob[g.w_off + (size_t)n * g.Ktot], g.Ktot);
            blob[ioff + n] = s16_pow2(-sc[n]);
        }
        for (int nt = 0; nt < g.N / 32; ++nt)
            for (int s = 0; s < ksteps; ++s)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int n = nt * 32 + (l & 31);
                        uint16_t t[3];
                        split16h(blob[g.w_off + (size_t)n * g.Ktot + s * 16 + 8 * (l >> 5) + e], sc[n], t);
                        const size_t base = (((size_t)nt * ksteps + s) * 2) * 512 + (size_t)l * 8 + e;
                        planes[base] = t[0];
                        planes[base + 512] = t[1];
                    }
        std::memcpy(&blob[off], planes.data(), planes.size() * 2);
        h->w6_of[g.w_off] = off;
        h->winv_of[g.w_off] = ioff;
    }
    void pack6(const PackedGemm& g) {
        if ((g.N % 64 && g.N % 96) || g.Ktot % 32 || h->w6_of.count(g.w_off)) return;
        if (use16()) return pack16(g);
        const size_t n_el = (size_t)g.N * g.Ktot;
        const size_t off = reserve((3 * n_el + 1) / 2);
        std::vector<uint16_t> planes(3 * n_el);
        const int ksteps = g.Ktot / 16;
        for (int nt = 0; nt < g.N / 32; ++nt)
            for (int s = 0; s < ksteps; ++s)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const float v = blob[g.w_off + (size_t)(nt * 32 + (l & 31)) * g.Ktot + s * 16 + 8 * (l >> 5) + e];
                        uint32_t b;
                        std::memcpy(&b, &v, 4);
                        const uint32_t bh = b & 0xffff0000u;
                        float fh;
                        std::memcpy(&fh, &bh, 4);
                        const float r1 = v - fh;
                        uint32_t b1;
                        std::memcpy(&b1, &r1, 4);
                        const uint32_t bm = b1 & 0xffff0000u;
                        float fm;
                        std::memcpy(&fm, &bm, 4);
                        const float r2 = r1 - fm;
                        uint32_t b2;
                        std::memcpy(&b2, &r2, 4);
                        const size_t base = (((size_t)nt * ksteps + s) * 3) * 512 + (size_t)l * 8 + e;
                        if (h->gemm_bf16) {   // opt-in bf16 mode: plane 0 = round-to-nearest-even(w), the others unused
                            planes[base] = bf16_rn(v);
                            continue;
                        }
                        planes[base] = (uint16_t)(bh >> 16);
                        planes[base + 512] = (uint16_t)(bm >> 16);
                        planes[base + 1024] = (uint16_t)(b2 >> 16);
                    }
        std::memcpy(&blob[off], planes.data(), planes.size() * 2);
        h->w6_of[g.w_off] = off;
    }
    // exact truncation split of one weight into three bf16 terms (tap_gemm6.h)
    static void split3h(float v, uint16_t (&o)[3]) {
        uint32_t b;
        std::memcpy(&b, &v, 4);
        const uint32_t bh = b & 0xffff0000u;
        float fh;
        std::memcpy(&fh, &bh, 4);
        const float r1 = v - fh;
        uint32_t b1;
        std::memcpy(&b1, &r1, 4);
        const uint32_t bm = b1 & 0xffff0000u;
        float fm;
        std::memcpy(&fm, &bm, 4);
        const float r2 = r1 - fm;
        uint32_t b2;
        std::memcpy(&b2, &r2, 4);
        o[0] = (uint16_t)(bh >> 16);
        o[1] = (uint16_t)(bm >> 16);
        o[2] = (uint16_t)(b2 >> 16);
    }
    // v_mfma_f32_16x16x32_bf16 operand fragments of a row-major [N][Ksrc] matrix in the blob:
    //   [n-tile of 16][k-step of 32][plane 3][lane 64][8 bf16],  lane (n = lane & 15, k = 8 * (lane >> 4) + e);
    // kmap[k'] = source column of padded column k', or -1 for a zero column
    size_t frag16(size_t src_off, int N, int Ksrc, const std::vector<int>& kmap, size_t* winv_off = nullptr) {
        const int ksteps = (int)kmap.size() / 32;
        const size_t n_el = (size_t)N * kmap.size();
        if (use16()) {   // split16.h: two fp16 planes of the scaled rows, [n-tile of 16][k-step of 32][plane 2][lane 64][8], + winv[N]
            const size_t off = reserve(n_el);
            const size_t ioff = reserve(N);
            std::vector<uint16_t> planes(2 * n_el);
            std::vector<int> sc(N);
            for (int n = 0; n < N; ++n) {
                sc[n] = row_scale(&blob[src_off + (size_t)n * Ksrc], Ksrc, &kmap);
                blob[ioff + n] = s16_pow2(-sc[n]);
            }
            for (int nt = 0; nt < N / 16; ++nt)
                for (int s = 0; s < ksteps; ++s)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int k = kmap[s * 32 + 8 * (l >> 4) + e], n = nt * 16 + (l & 15);
                            uint16_t t[3] = {0, 0, 0};
                            if (k >= 0) split16h(blob[src_off + (size_t)n * Ksrc + k], sc[n], t);
                            const size_t base = (((size_t)nt * ksteps + s) * 2) * 512 + (size_t)l * 8 + e;
                            planes[base] = t[0];
                            planes[base + 512] = t[1];
                        }
            std::memcpy(&blob[off], planes.data(), planes.size() * 2);
            if (winv_off) *winv_off = ioff;
            return off;
        }
        const size_t off = reserve((3 * n_el + 1) / 2);
        std::vector<uint16_t> planes(3 * n_el);
        for (int nt = 0; nt < N / 16; ++nt)
            for (int s = 0; s < ksteps; ++s)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int k = kmap[s * 32 + 8 * (l >> 4) + e];
                        uint16_t t[3] = {0, 0, 0};
                        if (k >= 0) {
                            if (h->gemm_bf16) t[0] = bf16_rn(blob[src_off + (size_t)(nt * 16 + (l & 15)) * Ksrc + k]);   // opt-in bf16 mode: one rounded plane
                            else split3h(blob[src_off + (size_t)(nt * 16 + (l & 15)) * Ksrc + k], t);
                        }
                        const size_t base = (((size_t)nt * ksteps + s) * 3) * 512 + (size_t)l * 8 + e;
                        planes[base] = t[0];
                        planes[base + 512] = t[1];
                        planes[base + 1024] = t[2];
                    }
        std::memcpy(&blob[off], planes.data(), planes.size() * 2);
        return off;
    }
    // thin_conv6.h image of a [64][128] layer
    void pack_t6(const PackedGemm& g) {
        if (g.N != 64 || g.Ktot != 128 || !g.has_bias || h->t6_of.count(g.w_off)) return;
        std::vector<int> km(128);
        for (int k = 0; k < 128; ++k) km[k] = k;
        size_t ioff = 0;
        h->t6_of[g.w_off] = frag16(g.w_off, 64, 128, km, &ioff);
        if (use16()) h->t6inv_of[g.w_off] = ioff;
    }
    // rb_fused6.h images of a residual block (k3 conv C -> C/2, then [1x1 over the hidden | optional shortcut over x])
    void rb6(ResBlockPlan& rb, bool sc) {
        const int C = rb.C, hid = C / 2;
        if ((C != 32 && C != 64 && C != 128) || rb.c3.N != hid || rb.c3.Ktot != 3 * C || rb.fused.N != C || rb.fused.Ktot != hid + (sc ? C : 0)) return;
        std::vector<int> k3(3 * C), kf;
        for (int k = 0; k < 3 * C; ++k) k3[k] = k;
        const int hcp = hid < 32 ? 32 : hid;
        for (int k = 0; k < hcp; ++k) kf.push_back(k < hid ? k : -1);
        for (int k = 0; sc && k < C; ++k) kf.push_back(hid + k);
        rb.w3f_off = frag16(rb.c3.w_off, hid, rb.c3.Ktot, k3, &rb.winv3_off);
        rb.wff_off = frag16(rb.fused.w_off, C, rb.fused.Ktot, kf, &rb.winvf_off);
        rb.hb0 = rb.hb1 = 0.f;
        for (int n = 0; n < hid; ++n) {
            double l1 = 0.0;
            for (int k = 0; k < rb.c3.Ktot; ++k) l1 += std::fabs((double)blob[rb.c3.w_off + (size_t)n * rb.c3.Ktot + k]);
            rb.hb1 = std::max(rb.hb1, (float)(l1 * 1.000001));
            rb.hb0 = std::max(rb.hb0, std::fabs(blob[rb.c3.b_off + n]));
        }
        rb.has6 = true;
    }
    // plain conv (stride 1 or k = 2*stride): packed[n][tap*cin + ci] = w[n][ci][tap]
    bool conv(const ConvSpec& s, PackedGemm& g, bool bias = true) {
        std::vector<float> w;
        if (!weight(s, w)) return false;
        const std::vector<float>* b = bias ? get(s.prefix + ".bias", s.cout) : nullptr;
        if (bias && !b) return false;
        g.has_bias = bias;
        g.N = s.cout;
        g.Ktot = s.k * s.cin;
        g.w_off = reserve((size_t)g.N * g.Ktot);
        for (int n = 0; n < s.cout; ++n)
            for (int ci = 0; ci < s.cin; ++ci)
                for (int t = 0; t < s.k; ++t)
                    blob[g.w_off + (size_t)n * g.Ktot + (size_t)t * s.cin + ci] = w[((size_t)n * s.cin + ci) * s.k + t];
        if (bias) {
            g.b_off = reserve(g.N);
            std::copy(b->begin(), b->end(), blob.begin() + g.b_off);
        }
        pack6(g);
        pack_t6(g);
        return true;
    }
    // transposed conv, k = 2*s: out row m = [x[m-1] | x[m]] * Wp,  n = p*cout + co,
    // Wp[n][j*cin + ci] = w[ci][co][p + (1-j)*s]
    bool convtr(const ConvSpec& s, PackedGemm& g) {
        std::vector<float> w;
        if (!weight(s, w)) return false;
        const std::vector<float>* b = get(s.prefix + ".bias", s.cout);
        if (!b) return false;
        g.N = s.s * s.cout;
        g.Ktot = 2 * s.cin;
        g.w_off = reserve((size_t)g.N * g.Ktot);
        for (int p = 0; p < s.s; ++p)
            for (int co = 0; co < s.cout; ++co)
                for (int j = 0; j < 2; ++j)
                    for (int ci = 0; ci < s.cin; ++ci)
                        blob[g.w_off + (size_t)(p * s.cout + co) * g.Ktot + (size_t)j * s.cin + ci] =
                            w[((size_t)ci * s.cout + co) * s.k + p + (1 - j) * s.s];
        g.b_off = reserve(g.N);
        for (int p = 0; p < s.s; ++p)
            for (int co = 0; co < s.cout; ++co) blob[g.b_off + (size_t)p * s.cout + co] = (*b)[co];
        pack6(g);
        pack_t6(g);
        return true;
    }
    bool resblock(const ConvSpec& c3, const ConvSpec& c1, const ConvSpec& sc, ResBlockPlan& rb) {
        rb.C = c3.cin;
        if (!conv(c3, rb.c3)) return false;
        std::vector<float> w1, ws;
        if (!weight(c1, w1) || !weight(sc, ws)) return false;
        const std::vector<float>* b1 = get(c1.prefix + ".bias", c1.cout);
        const std::vector<float>* bs = get(sc.prefix + ".bias", sc.cout);
        if (!b1 || !bs) return false;
        const int C = rb.C, hid = c1.cin;
        rb.fused.N = C;
        rb.fused.Ktot = hid + C;
        rb.fused.w_off = reserve((size_t)C * (hid + C));
        for (int n = 0; n < C; ++n) {
            for (int ci = 0; ci < hid; ++ci) blob[rb.fused.w_off + (size_t)n * (hid + C) + ci] = w1[(size_t)n * hid + ci];
            for (int ci = 0; ci < C; ++ci) blob[rb.fused.w_off + (size_t)n * (hid + C) + hid + ci] = ws[(size_t)n * C + ci];
        }
        rb.fused.b_off = reserve(C);
        for (int n = 0; n < C; ++n) blob[rb.fused.b_off + n] = (*b1)[n] + (*bs)[n];
        pack6(rb.fused);
        rb6(rb, true);
        return true;
    }
    // enc_front.h: constants of the bounds that stand in for the amax of the tensors inside a fused chain
    //   |stem(x)| <= sb0 + sb1 amax(x)            (largest |bias|, largest row 1-norm)
    //   |block out| <= fb0 + fb1h bound(hidden) + fb1x bound(block in)
    void chain_bounds(const PackedGemm& stem, const ResBlockPlan& rb, const PackedGemm& down, ac_handle::ChainBounds& cb) {
        cb.ok = false;
        if (!use16() || !rb.has6 || !rb.winv3_off || rb.C != 32 || stem.N != 32 || !h->t6inv_of.count(down.w_off)) return;
        const int hid = rb.C / 2;
        cb.sb0 = cb.sb1 = cb.fb0 = cb.fb1h = cb.fb1x = 0.f;
        for (int n = 0; n < stem.N; ++n) {
            double l1 = 0.0;
            for (int k = 0; k < stem.Ktot; ++k) l1 += std::fabs((double)blob[stem.w_off + (size_t)n * stem.Ktot + k]);
            cb.sb1 = std::max(cb.sb1, (float)(l1 * 1.000001));
            cb.sb0 = std::max(cb.sb0, std::fabs(blob[stem.b_off + n]));
        }
        for (int n = 0; n < rb.C; ++n) {
            double lh = 0.0, lx = 0.0;
            for (int k = 0; k < hid; ++k) lh += std::fabs((double)blob[rb.fused.w_off + (size_t)n * rb.fused.Ktot + k]);
            for (int k = hid; k < rb.fused.Ktot; ++k) lx += std::fabs((double)blob[rb.fused.w_off + (size_t)n * rb.fused.Ktot + k]);
            cb.fb1h = std::max(cb.fb1h, (float)(lh * 1.000001));
            cb.fb1x = std::max(cb.fb1x, (float)(lx * 1.000001));
            cb.fb0 = std::max(cb.fb0, std::fabs(blob[rb.fused.b_off + n]));
        }
        cb.ok = true;
    }
    // dec_tail.h: |transposed conv out| <= sb0 + sb1 amax(in)
    void tail_bounds(const PackedGemm& up, const ResBlockPlan& rb, const PackedGemm& head, ac_handle::ChainBounds& cb) {
        cb.ok = false;
        if (!use16() || !rb.has6 || !rb.winv3_off || rb.C != 32 || up.N != 64 || up.Ktot != 128 || !h->t6inv_of.count(up.w_off) || head.N != 1 || head.Ktot != 7 * 32) return;
        cb.sb0 = cb.sb1 = 0.f;
        for (int n = 0; n < up.N; ++n) {
            double l1 = 0.0;
            for (int k = 0; k < up.Ktot; ++k) l1 += std::fabs((double)blob[up.w_off + (size_t)n * up.Ktot + k]);
            cb.sb1 = std::max(cb.sb1, (float)(l1 * 1.000001));
            cb.sb0 = std::max(cb.sb0, std::fabs(blob[up.b_off + n]));
        }
        cb.ok = true;
    }
    bool lstm(const std::string& prefix, int D, int layers, LstmPlan& lp) {
        lp.D = D;
        lp.layers = layers;
        for (int l = 0; l < layers; ++l) {
            const std::string sfx = "_l" + std::to_string(l);
            const std::vector<float>* wih = get(prefix + ".weight_ih" + sfx, (size_t)4 * D * D);
            const std::vector<float>* whh = get(prefix + ".weight_hh" + sfx, (size_t)4 * D * D);
            const std::vector<float>* bih = get(prefix + ".bias_ih" + sfx, (size_t)4 * D);
            const std::vector<float>* bhh = get(prefix + ".bias_hh" + sfx, (size_t)4 * D);
            if (!wih || !whh || !bih || !bhh) return false;
            PackedGemm g;
            g.N = 4 * D;
            g.Ktot = D;
            g.w_off = reserve((size_t)4 * D * D);
            std::copy(wih->begin(), wih->end(), blob.begin() + g.w_off);
            g.b_off = reserve((size_t)4 * D);
            for (int n = 0; n < 4 * D; ++n) blob[g.b_off + n] = (*bih)[n] + (*bhh)[n];
            if (l == 0) pack6(g);
            lp.ih.push_back(g);
            // W_hh in MFMA B-fragment order: [ug][kstep][lane][u] = Whh[(lane&15 >> 2)*D + ug*4 + (lane&3)][kstep*16 + 4*(lane>>4) + u]
            const size_t off = reserve((size_t)4 * D * D);
            for (int ug = 0; ug < D / 4; ++ug)
                for (int ks = 0; ks < D / 16; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int u = 0; u < 4; ++u) {
                            const int j = lane & 15, kq = lane >> 4;
                            const int row = (j >> 2) * D + ug * 4 + (j & 3);
                            const int k = ks * 16 + 4 * kq + u;
                            blob[off + (((size_t)ug * (D / 16) + ks) * 64 + lane) * 4 + u] = (*whh)[(size_t)row * D + k];
                        }
            lp.hh_off.push_back(off);
            const size_t off2 = reserve((size_t)4 * D * D);
            for (int ug = 0; ug < D / 4; ++ug)
                for (int ks = 0; ks < D / 16; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int u = 0; u < 4; ++u) {
                            const int j = lane & 15, kq = lane >> 4;
                            const int row = (j >> 2) * D + ug * 4 + (j & 3);
                            const int k = ks * 16 + 4 * kq + u;
                            blob[off2 + (((size_t)ug * (D / 16) + ks) * 64 + lane) * 4 + u] = (*wih)[(size_t)row * D + k];
                        }
            lp.ihpk_off.push_back(off2);
        }
        if (D == LP_D && layers == 2) {
            // lstm_persist_kernel: [hh0, ih1, hh1][32 unit slices][4 waves = K quarters][4 gates][8 k-steps][64 lanes][4]
            const std::vector<float>* mats[3] = {get(prefix + ".weight_hh_l0", (size_t)4 * D * D), get(prefix + ".weight_ih_l1", (size_t)4 * D * D),
                                                 get(prefix + ".weight_hh_l1", (size_t)4 * D * D)};
            const size_t mat = (size_t)LP_SLICES * 4 * 4 * 8 * 256;
            lp.persist_off = reserve(3 * mat);
            for (int m = 0; m < 3; ++m)
                for (int idx = 0; idx < LP_SLICES; ++idx)
                    for (int w = 0; w < 4; ++w)
                        for (int n = 0; n < 4; ++n)
                            for (int ks = 0; ks < 8; ++ks)
                                for (int lane = 0; lane < 64; ++lane)
                                    for (int e = 0; e < 4; ++e)
                                        blob[lp.persist_off + m * mat + ((((size_t)idx * 4 + w) * 4 + n) * 8 + ks) * 256 + lane * 4 + e] =
                                            (*mats[m])[(size_t)(n * D + idx * 16 + (lane & 15)) * D + (w * 8 + ks) * 16 + 4 * (lane >> 4) + e];
            const std::vector<float>* mats6[4] = {mats[0], mats[1], mats[2], get(prefix + ".weight_ih_l0", (size_t)4 * D * D)};
            if (lstm16()) {
                // lstm_persist16_kernel: [hh0, ih1, hh1, ih0][32 slices][4 waves][4 gates][4 k-steps of 32][2 planes][64 lanes][8 fp16] of the
                // scaled rows; the two matrices of a layer share the accumulator, so their rows share the scale
                const size_t mat16 = (size_t)LP_SLICES * 4 * 4 * 4 * 2 * 512;
                lp.persist6_off = reserve((4 * mat16 + 1) / 2);
                lp.persist16_inv = reserve((size_t)2 * 4 * D);
                std::vector<int> sc((size_t)2 * 4 * D);
                const int layer_of[4] = {0, 1, 1, 0};
                for (int l = 0; l < 2; ++l)
                    for (int row = 0; row < 4 * D; ++row) {
                        uint32_t mx = 0;
                        for (int m = 0; m < 4; ++m)
                            if (layer_of[m] == l)
                                for (int k = 0; k < D; ++k) {
                                    uint32_t b;
                                    std::memcpy(&b, &(*mats6[m])[(size_t)row * D + k], 4);
                                    mx = std::max(mx, b & 0x7fffffffu);
                                }
                        sc[(size_t)l * 4 * D + row] = s16_exponent(mx, 40);
                        blob[lp.persist16_inv + (size_t)l * 4 * D + row] = s16_pow2(-sc[(size_t)l * 4 * D + row]);
                    }
                std::vector<uint16_t> pl16(4 * mat16);
                for (int m = 0; m < 4; ++m)
                    for (int idx = 0; idx < LP_SLICES; ++idx)
                        for (int w = 0; w < 4; ++w)
                            for (int n = 0; n < 4; ++n)
                                for (int ks = 0; ks < 4; ++ks)
                                    for (int lane = 0; lane < 64; ++lane)
                                        for (int e = 0; e < 8; ++e) {
                                            const int row = n * D + idx * 16 + (lane & 15);
                                            uint16_t t[3];
                                            split16h((*mats6[m])[(size_t)row * D + w * 128 + ks * 32 + 8 * (lane >> 4) + e], sc[(size_t)layer_of[m] * 4 * D + row], t);
                                            const size_t base = m * mat16 + (((((size_t)idx * 4 + w) * 4 + n) * 4 + ks) * 2) * 512 + (size_t)lane * 8 + e;
                                            pl16[base] = t[0];
                                            pl16[base + 512] = t[1];
                                        }
                std::memcpy(&blob[lp.persist6_off], pl16.data(), pl16.size() * 2);
            } else {
            // lstm_persist6_kernel: [hh0, ih1, hh1, ih0][32 slices][4 waves][4 gates][4 k-steps of 32][3 planes][64 lanes][8 bf16]
            const size_t mat6 = (size_t)LP_SLICES * 4 * 4 * 4 * 3 * 512;
            lp.persist6_off = reserve((4 * mat6 + 1) / 2);
            std::vector<uint16_t> pl6(4 * mat6);
            for (int m = 0; m < 4; ++m)
                for (int idx = 0; idx < LP_SLICES; ++idx)
                    for (int w = 0; w < 4; ++w)
                        for (int n = 0; n < 4; ++n)
                            for (int ks = 0; ks < 4; ++ks)
                                for (int lane = 0; lane < 64; ++lane)
                                    for (int e = 0; e < 8; ++e) {
                                        const float v = (*mats6[m])[(size_t)(n * D + idx * 16 + (lane & 15)) * D + w * 128 + ks * 32 + 8 * (lane >> 4) + e];
                                        uint32_t b;
                                        std::memcpy(&b, &v, 4);
                                        const uint32_t bh = b & 0xffff0000u;
                                        float fh;
                                        std::memcpy(&fh, &bh, 4);
                                        const float r1 = v - fh;
                                        uint32_t b1;
                                        std::memcpy(&b1, &r1, 4);
                                        const uint32_t bm = b1 & 0xffff0000u;
                                        float fm;
                                        std::memcpy(&fm, &bm, 4);
                                        const float r2 = r1 - fm;
                                        uint32_t b2;
                                        std::memcpy(&b2, &r2, 4);
                                        const size_t base = m * mat6 + (((((size_t)idx * 4 + w) * 4 + n) * 4 + ks) * 3) * 512 + (size_t)lane * 8 + e;
                                        pl6[base] = (uint16_t)(bh >> 16);
                                        pl6[base + 512] = (uint16_t)(bm >> 16);
                                        pl6[base + 1024] = (uint16_t)(b2 >> 16);
                                    }
            std::memcpy(&blob[lp.persist6_off], pl6.data(), pl6.size() * 2);
            }
            lp.has_persist = true;
        }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------
struct Act {          // a channels-last activation view
    const float* p;
    long long bs, ts;
    int L, C;
    const unsigned* amax = nullptr;   // split16.h: [B] largest-magnitude bits left by the producer (null: not reported).  A kernel that
                                      //   rewrites the tensor in place invalidates it: the caller must reset it (mimi_encoder_fwd does)
    int amax_n = 0;                   //   number of clips the slot was written for (a view of another batch shape must not use it)
};

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

struct ProfScope {
    ac_handle* h;
    hipStream_t st;
    ProfRec r{};
    bool on;
    ProfScope(ac_handle* h_, hipStream_t st_, const char* nm, double flops, double bytes, int count = 1)
        : h(h_), st(st_), on(h_->prof) {
        if (!on) return;
        r.name_id = prof_name(h, nm);
        r.count = count;
        r.flops = flops;
        r.bytes = bytes;
        r.e0 = next_event(h);
        r.e1 = next_event(h);
        (void)hipEventRecord(r.e0, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.e1, st);
        h->recs.push_back(r);
    }
};

// split16.h bookkeeping lives in the CALLER's workspace (include/audiocodecs_amd.h: "caller owns all device memory of a call"):
//   amax slots  [AMAX_SLOTS][pool_B clips][AMAX_STRIDE words], handed out in launch order
//   row ring    eight granules of `rows` words (row mode of the linear layers over merged token matrices)
// carve() binds the pool of the workspace at hand to the handle for the duration of the call; no entry point allocates,
// frees or synchronises.  AMAX_SLOTS bounds the producers of one pass (EnCodec ~30, WavTokenizer ~140, Mimi ~220, DAC ~110 per chunk).
constexpr int AMAX_SLOTS = 512;
inline size_t pool_bytes(int pool_B, size_t rows) {
    return (size_t)AMAX_SLOTS * std::max(pool_B, 1) * AMAX_STRIDE * 4 + 8 * align_up(std::max<size_t>(rows, 64), 64) * 4 + 256;
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

// start of a pass over B clips: all slots of the bound pool back to zero (one memset of AMAX_SLOTS x B lines on the caller's stream)
int amax_begin(ac_handle* h, hipStream_t st, int B) {
    h->amax_next = 0;
    if (!(h->split16 || h->gemm_bf16) || h->gemm_fp32 || !h->amax_buf) return AC_OK;
    if (B > h->amax_B) return fail(h, AC_ENOMEM, "workspace pool holds amax slots for %d clips, the pass has %d", h->amax_B, B);
    HIPCHK(h, hipMemsetAsync(h->amax_buf, 0, (size_t)AMAX_SLOTS * h->amax_B * AMAX_STRIDE * 4, st));
    return AC_OK;
}
// a fresh slot for a producer's output (null when the arithmetic does not use them)
unsigned* amax_new(ac_handle* h) {
    if (!(h->split16 || h->gemm_bf16) || h->gemm_fp32 || !h->amax_buf || h->amax_next >= AMAX_SLOTS) return nullptr;
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
    if (zero && hipMemsetAsync(r, 0, (size_t)rows * 4, st) != hipSuccess) return nullptr;
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
TapSeg make_seg(const Act& x, int s, int J, int pad /*PAD_**/, int extra, int kofs, const float* rel_len, int left = -1, int right = 0) {
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

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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
        std::snprintf(shape, sizeof shape, " B%d M%d N%d K%d J%d s%d", p.B, p.M, p.N, (int)kk, p.seg[0].J, p.seg[0].s);
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
        }
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
        if (h->gemm_bf16) TAP6_LAUNCH(WGM, WGN, WMT, WN, 1);                                                            \
        else if (p.winv) TAP6_LAUNCH(WGM, WGN, WMT, WN, 2);                                                             \
        else TAP6_LAUNCH(WGM, WGN, WMT, WN, 3);                                                                         \
    } while (0)
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
            static const char* force = std::getenv("AC_TAP_PICK");     // developer override: 0 / 1 / 2
            if (force && force[0] >= '0' && force[0] <= '2') pick = force[0] - '0';
        }
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

// What a layer should produce: the raw output, its ELU, or both (SEANet consumers all start with ELU;
// shortcuts, LSTMs and the public outputs want the raw value).
struct Out {
    float* raw = nullptr;
    float* elu = nullptr;
};
struct Act2 {       // a layer output in up to two flavours (same shape/strides)
    Act raw{}, elu{};
};

// the [64][128] layers on 64-float super-rows (thin_conv6.h); returns 1 when the shape does not qualify
int try_thin6(ac_handle* h, hipStream_t st, const PackedGemm& g, const Act& x, int width, int edge, Out out, int B, const unsigned** amax_out = nullptr) {
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
    ProfScope ps(h, st, h->gemm_bf16 ? "thin_conv6_kernel<1>" : s16 ? "thin_conv6_kernel<2>" : "thin_conv6_kernel<3>", 2.0 * B * p.M * 64.0 * 128.0,
                 (double)B * p.M * 256.0 * (1 + (out.raw ? 1 : 0) + (out.elu ? 1 : 0)));
    if (h->gemm_bf16) hipLaunchKernelGGL(thin_conv6_kernel<1>, dim3(grid), dim3(256), T6_LDS, st, p);
    else if (s16) hipLaunchKernelGGL(thin_conv6_kernel<2>, dim3(grid), dim3(256), T6_LDS, st, p);
    else hipLaunchKernelGGL(thin_conv6_kernel<3>, dim3(grid), dim3(256), T6_LDS, st, p);
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
    if (!h->split16 || h->gemm_bf16 || h->gemm_fp32 || !rb.winv3_off) return false;
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
int launch_rb_fused6(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, Out out, int B, int pad = PAD_REFLECT, const unsigned** amax_out = nullptr) {
    using Cfg = Rb6Cfg<C, SC>;
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
    if (const char* d = std::getenv("AC_RB6_DBG")) p.dbg = std::atoi(d);
    const bool s16 = rb_split16(h, st, rb, x, B, p, amax_out);
    if (s16 && !p.amax_in) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
    const size_t lds6 = s16 ? Cfg::lds_bytes16 : Cfg::lds_bytes;
    if (int rc = ensure_lds(h, h->gemm_bf16 ? reinterpret_cast<const void*>(rb_fused6_kernel<C, SC, 1>) : s16 ? reinterpret_cast<const void*>(rb_fused6_kernel<C, SC, 2>) : reinterpret_cast<const void*>(rb_fused6_kernel<C, SC, 3>), lds6)) return rc;
    const long long total = (long long)B * p.ntiles;
    const int per_cu = s16 ? rb6_occupancy<C, SC, 2>() : rb6_occupancy<C, SC, 3>();
    const int grid = (int)std::min<long long>(total, (long long)per_cu * 256);   // persistent
    const double L = x.raw.L;
    const std::string np6 = h->gemm_bf16 ? ", 1>" : s16 ? ", 2>" : ", 3>";
    ProfScope ps(h, st, ((!SC ? "rb_fused6_kernel<64, false" : C == 32 ? "rb_fused6_kernel<32, true" : "rb_fused6_kernel<64, true") + np6).c_str(),
                 2.0 * B * L * ((double)(C / 2) * 3 * C + (double)C * (C / 2 + (SC ? C : 0))),
                 (double)B * L * C * 4.0 * (1 + (out.raw ? 1 : 0) + (out.elu ? 1 : 0)));
    if (h->gemm_bf16) hipLaunchKernelGGL((rb_fused6_kernel<C, SC, 1>), dim3(grid), dim3(256), Cfg::lds_bytes, st, p);
    else if (s16) hipLaunchKernelGGL((rb_fused6_kernel<C, SC, 2>), dim3(grid), dim3(256), lds6, st, p);
    else hipLaunchKernelGGL((rb_fused6_kernel<C, SC, 3>), dim3(grid), dim3(256), Cfg::lds_bytes, st, p);
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
    const bool s16 = rb_split16(h, st, rb, x, B, p, amax_out);
    if (s16 && !p.amax_in) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
    if (int rc = ensure_lds(h, h->gemm_bf16 ? reinterpret_cast<const void*>(rb128_fused6_kernel<SC, 1>) : s16 ? reinterpret_cast<const void*>(rb128_fused6_kernel<SC, 2>) : reinterpret_cast<const void*>(rb128_fused6_kernel<SC, 3>), Cfg::lds_bytes)) return rc;
    const long long total = (long long)B * p.ntiles;
    const int grid = (int)std::min<long long>(total, 256);   // persistent, one workgroup per CU
    const double L = x.raw.L;
    const std::string np6 = h->gemm_bf16 ? ", 1>" : s16 ? ", 2>" : ", 3>";
    ProfScope ps(h, st, ((SC ? "rb128_fused6_kernel<true" : "rb128_fused6_kernel<false") + np6).c_str(),
                 2.0 * B * L * (64.0 * 384 + 128.0 * (64 + (SC ? 128 : 0))),
                 (double)B * L * 128 * 4.0 * (1 + (out.raw ? 1 : 0) + (out.elu ? 1 : 0)));
    if (h->gemm_bf16) hipLaunchKernelGGL((rb128_fused6_kernel<SC, 1>), dim3(grid), dim3(512), Cfg::lds_bytes, st, p);
    else if (s16) hipLaunchKernelGGL((rb128_fused6_kernel<SC, 2>), dim3(grid), dim3(512), Cfg::lds_bytes, st, p);
    else hipLaunchKernelGGL((rb128_fused6_kernel<SC, 3>), dim3(grid), dim3(512), Cfg::lds_bytes, st, p);
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
              Out out, Act2* y, int padl = -1, const float* alpha = nullptr, const float* alpha_inv = nullptr, int Lp = -1) {
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

int thin_head(ac_handle* h, hipStream_t st, const PackedGemm& g, int F, int k, int pad, const Act& x, int B, float* sig, int padl = -1,
              int tanh_out = 0) {
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
    return h->fuse_chains && h->arch == ARCH_ENCODEC && !h->noncausal && h->split16 && !h->gemm_bf16 && !h->gemm_fp32 && h->enc_front.ok &&
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
    if (const char* sc = std::getenv("AC_FRONT_SEG")) p.seg_chunks = std::max(1, std::atoi(sc));
    p.segs_per_clip = cdiv(nchunks, p.seg_chunks);
    // amax of the samples (one read of 4 B per sample; as 16-byte vectors where the clip pitch allows)
    const bool v4 = T % 4 == 0 && aligned16(sig);
    p.amax_sig = amax_of(h, st, sig, T, v4 ? 4 : 1, v4 ? T / 4 : T, v4 ? 4 : 1, B, nullptr);
    if (!p.amax_sig) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
    p.amax_out = amax_new(h);
    p.sb0 = h->enc_front.sb0; p.sb1 = h->enc_front.sb1;
    p.hb0 = rb.hb0; p.hb1 = rb.hb1;
    p.fb0 = h->enc_front.fb0; p.fb1h = h->enc_front.fb1h; p.fb1x = h->enc_front.fb1x;
    size_t lds = EF_LDS;
    if (const char* lp = std::getenv("AC_FRONT_LDSPAD")) lds += (size_t)std::atoi(lp);     // developer: force one workgroup per CU
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
    return h->fuse_chains && h->arch == ARCH_ENCODEC && !h->noncausal && h->split16 && !h->gemm_bf16 && !h->gemm_fp32 && h->dec_tail.ok &&
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
    if (const char* sc = std::getenv("AC_TAIL_SEG")) p.seg_chunks = std::max(1, std::atoi(sc));
    p.segs_per_clip = cdiv(nchunks, p.seg_chunks);
    p.amax_x = amax_of(h, st, xe.p, xe.bs, xe.ts, xe.L, xe.C, B, xe.amax_n == B ? xe.amax : nullptr);
    if (!p.amax_x) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
    p.ub0 = h->dec_tail.sb0; p.ub1 = h->dec_tail.sb1;
    p.hb0 = rb.hb0; p.hb1 = rb.hb1;
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

struct LstmWs {
    float *gin, *gin1, *hseq0, *hseq1, *c;   // c holds one [B][D] cell state per layer
};

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
    static const bool fuse_env = !(std::getenv("AC_LSTM_FUSE_IN") && std::getenv("AC_LSTM_FUSE_IN")[0] == '0');
    const bool fuse_in = persist && !h->gemm_fp32 && fuse_env && aligned16(x.p) && x.bs % 4 == 0;   // lstm_persist6.h computes W_ih0 * x[t] itself
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
        ProfScope ps(h, st, h->gemm_fp32 ? "lstm_persist_kernel" : lp.persist16_inv ? (fuse_in ? "lstm_persist16_kernel<true>" : "lstm_persist16_kernel<false>") : "lstm_persist6_kernel<3>", 2.0 * T * (double)B * 4 * D * D * (nroles + (fuse_in ? 1 : 0)),
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
            const char* ld = std::getenv("AC_LSTM_DBG");
            q.dbg = ld ? std::atoi(ld) : 0;
            if (!h->gemm_fp32) {   // split-operand products on the bf16 pipe (lstm_persist6.h): h travels as bf16 plane blocks
                LstmPersist6Params q6{};
                q6.base = q;
                const bool l16 = lp.persist16_inv != 0;     // split16.h planes
                q6.base.h_ts = (long long)((B + 31) / 32 * 2) * (l16 ? LP16_GROUP_BYTES : LP6_GROUP_BYTES);
                if (l16) {
                    q6.winv = h->blob + lp.persist16_inv;
                    q6.amax_x = fuse_in ? x_amax : nullptr;
                }
                q6.w_pk6 = reinterpret_cast<const __bf16*>(h->blob + lp.persist6_off);
                q6.bias0 = h->blob + lp.ih[0].b_off;
                q6.fuse_in = fuse_in ? 1 : 0;
                q6.poison = poison;
                q6.hseq0_local = ws.gin1;     // free on this path (the per-step kernels' layer-1 pre-activations)
                HIPCHK(h, hipMemsetAsync(h->lp_ctl, 0, LP_CTL_WORDS * sizeof(unsigned), st));
                // the exchange validates itself: every element of the h buffers starts as the "not yet written" pattern
                const size_t hbytes = (size_t)T * (size_t)q6.base.h_ts;
                HIPCHK(h, hipMemsetAsync(ws.hseq0, 0xFF, hbytes, st));
                HIPCHK(h, hipMemsetAsync(ws.hseq1, 0xFF, hbytes, st));
                if (l16) HIPCHK(h, hipMemsetAsync(ws.gin1, 0xFF, hbytes, st));
                void* args6[] = {&q6};
                const void* kfn = l16 ? (fuse_in ? reinterpret_cast<const void*>(lstm_persist16_kernel<true>) : reinterpret_cast<const void*>(lstm_persist16_kernel<false>))
                                      : reinterpret_cast<const void*>(lstm_persist6_kernel<3>);
                HIPCHK(h, hipLaunchCooperativeKernel(kfn, dim3(256), dim3(l16 ? 512 : 256), args6, 0, st));
                tail(c0, q.B, poison);
                if (q.dbg & 32) {   // developer trace: 100 MHz real-time stamps of steps 100 .. 103 (lstm_persist6.h)
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
                        std::fprintf(stderr, "\n");
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
    static const bool exact_env = std::getenv("AC_RVQ") && std::strcmp(std::getenv("AC_RVQ"), "fp32") == 0;   // developer A/B switch
    if (h->cb16 && HV == 8 && !exact_env) {   // split16 products on the fp16 matrix pipe (rvq16.h)
        RvqEnc16Params q{};
        q.base = p;
        q.epk16 = reinterpret_cast<const _Float16*>(h->blob + h->cb16);
        q.einv = h->blob + h->cb16_inv;
        ProfScope ps(h, st, "rvq_encode16_kernel", 2.0 * F * (double)p.C * p.H * K,
                     (double)F * p.H * 4 + (double)F * K * 8 + (double)K * p.C * p.H * 4);
        if (MS == 3) hipLaunchKernelGGL((rvq_encode16_kernel<8, 3, false>), grid, block, 0, st, q);
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

// ---------------------------------------------------------------------------------------------
// workspace layout
// ---------------------------------------------------------------------------------------------
constexpr int NACT = 6;   // rotating activation buffers: x.raw, x.elu, hidden, y.elu (+ y.raw when capturing)

struct Workspace {
    size_t act_floats = 0;     // each of the NACT rotating activation buffers
    size_t gin = 0, hseq = 0, c = 0;
    int pool_B = 0;            // split16.h pool (amax slots for pool_B clips + row ring for pool_rows rows), at the head of the workspace
    size_t pool_rows = 0;
    size_t total_bytes = 0;
};
// every planner ends here: the pool is part of what ac_*_workspace_bytes reports
inline void add_pool(Workspace& w, int pool_B, size_t rows) {
    w.pool_B = pool_B;
    w.pool_rows = rows;
    w.total_bytes += pool_bytes(pool_B, rows);
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
    w.hseq = align_up((size_t)N * ((B + 31) / 32 * 32) * h->D * 3 / 2, 64);   // clips padded to the 32-clip workgroup tile; x1.5: bf16 plane blocks (lstm_persist6.h)
    w.gin = std::max(align_up((size_t)N * B * 4 * h->D, 64), w.hseq);          // gin1 doubles as layer 0's local h copy on the persistent path
    w.c = align_up((size_t)2 * B * h->D, 64);
    w.total_bytes = (NACT * w.act_floats + 2 * w.gin + 2 * w.hseq + w.c) * sizeof(float) + 256;
    add_pool(w, B, 0);
    return w;
}

struct WsPtrs {
    float* act[NACT];
    bool used[NACT];
    LstmWs lstm;
    float* take() {
        for (int i = 0; i < NACT; ++i)
            if (!used[i]) { used[i] = true; return act[i]; }
        return nullptr;   // cannot happen: at most 5 are live at once
    }
    void give(const float* p) {
        for (int i = 0; i < NACT; ++i)
            if (act[i] == p) used[i] = false;
    }
    void give(const Act2& a) { give(a.raw.p); give(a.elu.p); }
};

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
        volatile unsigned* s = h->sticky;
        if (s[ST_LSTM_TIMEOUT] || s[ST_LSTM_PLACEMENT]) {
            const unsigned a = s[ST_LSTM_TIMEOUT], b = s[ST_LSTM_PLACEMENT];
            s[ST_LSTM_TIMEOUT] = 0;
            s[ST_LSTM_PLACEMENT] = 0;
            h->lstm_step_only = true;   // self-heal: the per-step kernels need no co-residency
            return fail(h, AC_EHIP,
                        "an EARLIER call's persistent LSTM launch failed (%u bounded waits expired, %u launches without 32 workgroups on "
                        "every XCD -- is the GPU shared?): that call's outputs were set to NaN; the handle now uses the per-step LSTM "
                        "kernels, repeat the call", a, b);
        }
        if (s[ST_BAD_TOKEN]) {
            const unsigned a = s[ST_BAD_TOKEN];
            s[ST_BAD_TOKEN] = 0;
            return fail(h, AC_EINVAL, "an EARLIER ac_decode / ac_dequantize call got %u token ids outside [0, codebook_size): those frames were set to NaN", a);
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

#include "mimi_path.h"
#include "dac_path.h"
#include "wavtok_path.h"

}  // namespace

// ---------------------------------------------------------------------------------------------
// exported entry points
// ---------------------------------------------------------------------------------------------
extern "C" {

int ac_version(void) { return 310; }   // 310: split16 arithmetic (AC_PRECISION_FP32_BF16X3 beside it), ac_debug_split_row

int ac_create(const ac_config* cfg, ac_handle** out) {
    if (!cfg || !out) return AC_EINVAL;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(ac_config)) return AC_EINVAL;
    if (cfg->num_ratios < 1 || cfg->num_ratios > AC_MAX_RATIOS || cfg->num_filters < 1 || cfg->hidden_size < 16 ||
        cfg->hidden_size % 16 || cfg->compress < 1 || cfg->num_lstm_layers < 1 || cfg->num_lstm_layers > 2 || cfg->codebook_size % 32 ||
        cfg->codebook_size < 16 || cfg->num_quantizers < 1 || cfg->kernel_size < 1 || cfg->kernel_size > 8 ||
        cfg->last_kernel_size < 1 || cfg->last_kernel_size > 8 || cfg->residual_kernel_size < 1 || cfg->residual_kernel_size > 8)
        return AC_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device || cfg->device < 0) return AC_ENODEV;
    ac_handle* h = new (std::nothrow) ac_handle();
    if (!h) return AC_ENOMEM;
    h->cfg = *cfg;
    h->hop = 1;
    for (int i = 0; i < cfg->num_ratios; ++i) h->hop *= cfg->upsampling_ratios[i];
    h->D = cfg->num_filters << cfg->num_ratios;
    *out = h;
    return AC_OK;
}

int ac_mimi_create(const ac_mimi_config* cfg, ac_handle** out) {
    if (!cfg || !out) return AC_EINVAL;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(ac_mimi_config)) return AC_EINVAL;
    const ac_mimi_config& c = *cfg;
    if (c.num_ratios < 1 || c.num_ratios > AC_MAX_RATIOS || c.num_filters < 1 || c.hidden_size < 16 || c.hidden_size % 16 ||
        c.hidden_size > 64 * LN_MAXV || c.compress < 1 || c.codebook_size % 32 || c.codebook_size < 32 || c.codebook_dim < 16 ||
        c.codebook_dim % 16 || c.codebook_dim > 256 || c.num_quantizers < 1 || c.num_semantic_quantizers < 1 ||
        c.num_semantic_quantizers > c.num_quantizers || c.kernel_size < 1 || c.kernel_size > 8 || c.last_kernel_size < 1 ||
        c.last_kernel_size > 8 || c.residual_kernel_size < 1 || c.residual_kernel_size > 8 || c.num_hidden_layers < 0 ||
        c.num_attention_heads < 1 || (c.head_dim != 16 && c.head_dim != 32 && c.head_dim != 64) || c.intermediate_size < 16 ||
        c.intermediate_size % 4 || c.sliding_window < 1 || c.resample_stride < 1 || c.resample_stride > 4 || !(c.norm_eps > 0.f) ||
        !(c.rope_theta > 0.f))
        return AC_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= c.device || c.device < 0) return AC_ENODEV;
    ac_handle* h = new (std::nothrow) ac_handle();
    if (!h) return AC_ENOMEM;
    h->arch = ARCH_MIMI;
    h->mcfg = c;
    h->hop = c.resample_stride;
    for (int i = 0; i < c.num_ratios; ++i) h->hop *= c.upsampling_ratios[i];
    h->D = c.num_filters << c.num_ratios;
    h->mimi.D = h->D;
    *out = h;
    return AC_OK;
}

int ac_dac_create(const ac_dac_config* cfg, ac_handle** out) {
    if (!cfg || !out) return AC_EINVAL;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(ac_dac_config)) return AC_EINVAL;
    const ac_dac_config& c = *cfg;
    if (c.num_ratios < 1 || c.num_ratios > AC_MAX_RATIOS || c.encoder_hidden_size < 1 || c.decoder_hidden_size < (1 << c.num_ratios) ||
        c.decoder_hidden_size % (1 << c.num_ratios) || c.n_codebooks < 1 || c.codebook_size % 64 || c.codebook_size < 64 ||
        c.codebook_dim != DAC_D || c.num_dilations < 1 || c.num_dilations > AC_MAX_DILATIONS)
        return AC_EINVAL;
    const int H = c.encoder_hidden_size << c.num_ratios;
    if (H % 64 || H > 1024) return AC_EINVAL;
    for (int i = 0; i < c.num_dilations; ++i)
        if (c.dilations[i] < 1 || 6 * c.dilations[i] > GEN_EXTRA) return AC_EINVAL;
    for (int i = 0; i < c.num_ratios; ++i)
        if (c.downsampling_ratios[i] < 1 || c.upsampling_ratios[i] < 1) return AC_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= c.device || c.device < 0) return AC_ENODEV;
    ac_handle* h = new (std::nothrow) ac_handle();
    if (!h) return AC_ENOMEM;
    h->arch = ARCH_DAC;
    h->dcfg = c;
    h->hop = 1;
    for (int i = 0; i < c.num_ratios; ++i) h->hop *= c.downsampling_ratios[i];
    h->D = H;
    h->dac.H = H;
    *out = h;
    return AC_OK;
}

int ac_wavtok_create(const ac_wavtok_config* cfg, ac_handle** out) {
    if (!cfg || !out) return AC_EINVAL;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(ac_wavtok_config)) return AC_EINVAL;
    const ac_wavtok_config& c = *cfg;
    if (c.num_ratios < 1 || c.num_ratios > AC_MAX_RATIOS || c.num_filters < 1 || c.dimension < 16 || c.dimension % 16 || c.dimension > 512 ||
        c.compress < 1 || c.num_lstm_layers < 1 || c.num_lstm_layers > 2 || c.codebook_size % 32 || c.codebook_size < 32 ||
        c.kernel_size < 1 || c.kernel_size > 8 || c.last_kernel_size < 1 || c.last_kernel_size > 8 || c.residual_kernel_size < 1 ||
        c.residual_kernel_size > 8 || (c.backbone_dim != 256 && c.backbone_dim != 768) || c.intermediate_dim < 16 || c.intermediate_dim % 4 ||
        c.num_layers < 0 || c.adanorm_num_embeddings < 1 || c.bandwidth_id < 0 || c.bandwidth_id >= c.adanorm_num_embeddings ||
        c.num_groups < GN_GPW || c.num_groups % GN_GPW || c.backbone_dim % c.num_groups || (c.backbone_dim / c.num_groups * GN_GPW) % 4 ||
        c.backbone_dim / c.num_groups * GN_GPW > 1024 || c.n_fft < 4 || c.n_fft % 2)
        return AC_EINVAL;
    int hop = 1;
    for (int i = 0; i < c.num_ratios; ++i) {
        if (c.ratios[i] < 1) return AC_EINVAL;
        hop *= c.ratios[i];
    }
    // the inverse STFT is a GEMM over n_fft / hop whole frames per output row; "same" padding trims (n_fft - hop) / 2
    if (c.n_fft % hop || c.n_fft / hop < 1 || c.n_fft / hop > 8 || hop % 4 || ((c.n_fft - hop) / 2) % 4 || (c.n_fft - hop) % 2) return AC_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= c.device || c.device < 0) return AC_ENODEV;
    ac_handle* h = new (std::nothrow) ac_handle();
    if (!h) return AC_ENOMEM;
    h->arch = ARCH_WAVTOK;
    h->wcfg = c;
    h->noncausal = true;
    // the SEANet encoder is the EnCodec encoder plan with centred padding: describe it in h->cfg
    ac_config& e = h->cfg;
    e.struct_size = (int32_t)sizeof(ac_config);
    e.sampling_rate = c.sampling_rate;
    e.num_filters = c.num_filters;
    e.hidden_size = c.dimension;
    e.num_ratios = c.num_ratios;
    for (int i = 0; i < c.num_ratios; ++i) e.upsampling_ratios[i] = c.ratios[i];
    e.kernel_size = c.kernel_size;
    e.last_kernel_size = c.last_kernel_size;
    e.residual_kernel_size = c.residual_kernel_size;
    e.compress = c.compress;
    e.num_lstm_layers = c.num_lstm_layers;
    e.codebook_size = c.codebook_size;
    e.num_quantizers = 1;
    e.device = c.device;
    h->hop = hop;
    h->D = c.num_filters << c.num_ratios;
    *out = h;
    return AC_OK;
}

// WavTokenizer checkpoints spell the encoder's modules the encodec-library way; the packer knows the HF spelling
static std::string wavtok_key(const std::string& name) {
    static const std::string pre = "feature_extractor.encodec.encoder.model.";
    if (name.compare(0, pre.size(), pre) != 0) return name;
    std::string k = "encoder.layers." + name.substr(pre.size());
    const size_t cc = k.find(".conv.conv.");
    if (cc != std::string::npos) k = k.substr(0, cc) + ".conv." + k.substr(cc + 11);
    auto ends = [&](const char* sfx) { const size_t n = std::strlen(sfx); return k.size() >= n && k.compare(k.size() - n, n, sfx) == 0; };
    if (ends(".weight_g")) k = k.substr(0, k.size() - 9) + ".parametrizations.weight.original0";
    else if (ends(".weight_v")) k = k.substr(0, k.size() - 9) + ".parametrizations.weight.original1";
    return k;
}

int ac_load_weights(ac_handle* h, const char* name, const void* host_ptr, size_t bytes) {
    if (!h || !name || !host_ptr) return h ? fail(h, AC_EINVAL, "null argument") : AC_EINVAL;
    if (h->finalized) return fail(h, AC_ESTATE, "handle already finalized");
    if (bytes % 4) return fail(h, AC_EINVAL, "tensor '%s': byte size %zu is not a multiple of 4", name, bytes);
    const float* f = static_cast<const float*>(host_ptr);
    h->host[h->arch == ARCH_WAVTOK ? wavtok_key(name) : std::string(name)].assign(f, f + bytes / 4);
    return AC_OK;
}

static int upload_blob(ac_handle* h, Packer& pk, int device) {
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

int ac_set_precision(ac_handle* h, int precision) {
    if (!h) return AC_EINVAL;
    if (h->finalized) return fail(h, AC_ESTATE, "ac_set_precision must precede ac_finalize (weights are packed for one arithmetic)");
    if (precision < AC_PRECISION_FP32 || precision > AC_PRECISION_FP32_BF16X3) return fail(h, AC_EINVAL, "unknown precision %d", precision);
    h->precision = precision;
    return AC_OK;
}

int ac_finalize(ac_handle* h) {
    if (!h) return AC_EINVAL;
    if (h->finalized) return fail(h, AC_ESTATE, "handle already finalized");
    {   // arithmetic of the GEMM-shaped kernels: ac_set_precision, else the environment variable AC_GEMM (fp32 | bf16)
        int pr = h->precision;
        if (pr < 0) {
            const char* gm = std::getenv("AC_GEMM");
            pr = gm && std::strcmp(gm, "fp32") == 0 ? AC_PRECISION_FP32_EXACT : gm && std::strcmp(gm, "bf16") == 0 ? AC_PRECISION_BF16 :
                 gm && std::strcmp(gm, "bf16x3") == 0 ? AC_PRECISION_FP32_BF16X3 : AC_PRECISION_FP32;
        }
        const char* fz = std::getenv("AC_FUSE");
        h->fuse_chains = !(fz && fz[0] == '0');
        h->gemm_fp32 = pr == AC_PRECISION_FP32_EXACT;
        h->gemm_bf16 = pr == AC_PRECISION_BF16;
        h->split16 = pr == AC_PRECISION_FP32;
    }
    if (h->arch == ARCH_MIMI) {
        Packer pk{h};
        if (int rc = mimi_finalize(h, pk)) return rc;
        return upload_blob(h, pk, h->mcfg.device);
    }
    if (h->arch == ARCH_DAC) {
        Packer pk{h};
        if (int rc = dac_finalize(h, pk)) return rc;
        return upload_blob(h, pk, h->dcfg.device);
    }
    if (h->arch == ARCH_WAVTOK) {
        Packer pk{h};
        if (int rc = wavtok_finalize(h, pk)) return rc;
        return upload_blob(h, pk, h->wcfg.device);
    }
    const ac_config& c = h->cfg;
    Arch a = make_arch(c);
    Packer pk{h};
    // a caller in encode-only / decode-only mode hands over just that half (encodec.py:67-71 deletes the other one)
    h->has_enc = h->has_dec = false;
    for (const auto& kv : h->host) {
        if (kv.first.compare(0, 8, "encoder.") == 0) h->has_enc = true;
        if (kv.first.compare(0, 8, "decoder.") == 0) h->has_dec = true;
    }
    if (!h->has_enc && !h->has_dec) return fail(h, AC_ESTATE, "no encoder.* or decoder.* tensor was loaded");
    bool ok = true;
    h->enc_rb.resize(c.num_ratios);
    h->enc_down.resize(c.num_ratios);
    h->dec_up.resize(c.num_ratios);
    h->dec_rb.resize(c.num_ratios);
    if (h->has_enc) {
        ok = pk.conv(a.enc_stem, h->enc_stem);
        for (int i = 0; ok && i < c.num_ratios; ++i) {
            ok = ok && pk.resblock(a.enc_rb3[i], a.enc_rb1[i], a.enc_rbs[i], h->enc_rb[i]);
            ok = ok && pk.conv(a.enc_down[i], h->enc_down[i]);
        }
        ok = ok && pk.lstm(a.enc_lstm, a.D, c.num_lstm_layers, h->enc_lstm);
        ok = ok && pk.conv(a.enc_final, h->enc_final);
        if (ok) pk.chain_bounds(h->enc_stem, h->enc_rb[0], h->enc_down[0], h->enc_front);
    }
    if (h->has_dec) {
        ok = ok && pk.conv(a.dec_first, h->dec_first);
        ok = ok && pk.lstm(a.dec_lstm, a.D, c.num_lstm_layers, h->dec_lstm);
        for (int i = 0; ok && i < c.num_ratios; ++i) {
            ok = ok && pk.convtr(a.dec_up[i], h->dec_up[i]);
            ok = ok && pk.resblock(a.dec_rb3[i], a.dec_rb1[i], a.dec_rbs[i], h->dec_rb[i]);
        }
        ok = ok && pk.conv(a.dec_head, h->dec_head);
        if (ok) pk.tail_bounds(h->dec_up[c.num_ratios - 1], h->dec_rb[c.num_ratios - 1], h->dec_head, h->dec_tail);
    }
    if (!ok) return pk.rc;
    // codebooks: plain [K][C][H], MFMA B-fragment order, squared norms
    const int C = c.codebook_size, H = c.hidden_size, Q = c.num_quantizers;
    h->cb_plain = pk.reserve((size_t)Q * C * H);
    h->cb_packed = pk.reserve((size_t)Q * C * H);
    h->cb_ee = pk.reserve((size_t)Q * C);
    for (int q = 0; q < Q; ++q) {
        const std::vector<float>* e = pk.get("quantizer.layers." + std::to_string(q) + ".codebook.embed", (size_t)C * H);
        if (!e) return pk.rc;
        std::copy(e->begin(), e->end(), pk.blob.begin() + h->cb_plain + (size_t)q * C * H);
        for (int code = 0; code < C; ++code) {
            double ss = 0.0;
            for (int d = 0; d < H; ++d) ss += (double)(*e)[(size_t)code * H + d] * (*e)[(size_t)code * H + d];
            pk.blob[h->cb_ee + (size_t)q * C + code] = (float)ss;
        }
        const int HV = H / 16;
        for (int ct = 0; ct < C / 16; ++ct)
            for (int v = 0; v < HV; ++v)
                for (int lane = 0; lane < 64; ++lane)
                    for (int u = 0; u < 4; ++u)
                        pk.blob[h->cb_packed + (size_t)q * C * H + (((size_t)ct * HV + v) * 64 + lane) * 4 + u] =
                            (*e)[(size_t)(ct * 16 + (lane & 15)) * H + v * 16 + 4 * (lane >> 4) + u];
    }
    if (pk.use16() && H % 32 == 0 && H / 16 == 8) {
        // rvq16.h: split16 image of every table, one power-of-two scale per table, halves in the lanes' own dim order:
        //   [q][code tile 16][k-step s of 32][plane 2][lane (j, kq)][e 8]  <->  dim 16 (2s + e/4) + 4 kq + e%4 of code 16 ct + j
        const int KS = H / 32;
        h->cb16 = pk.reserve((size_t)Q * C * H);                  // 2 planes x 2 bytes = 4 bytes per element
        h->cb16_inv = pk.reserve((size_t)Q);
        std::vector<uint16_t> img((size_t)Q * C * H * 2);
        for (int q = 0; q < Q; ++q) {
            const float* e = &pk.blob[h->cb_plain + (size_t)q * C * H];
            const int se = Packer::row_scale(e, (size_t)C * H);
            pk.blob[h->cb16_inv + q] = s16_pow2(-se);
            for (int ct = 0; ct < C / 16; ++ct)
                for (int s = 0; s < KS; ++s)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e8 = 0; e8 < 8; ++e8) {
                            const int j = lane & 15, kq = lane >> 4;
                            const int dim = 16 * (2 * s + e8 / 4) + 4 * kq + e8 % 4;
                            uint16_t t[3];
                            Packer::split16h(e[(size_t)(ct * 16 + j) * H + dim], se, t);
                            const size_t base = ((((size_t)q * (C / 16) + ct) * KS + s) * 2) * 512 + (size_t)lane * 8 + e8;
                            img[base] = t[0];
                            img[base + 512] = t[1];
                        }
        }
        std::memcpy(&pk.blob[h->cb16], img.data(), img.size() * 2);
    }
    return upload_blob(h, pk, c.device);
}

int ac_num_frames(const ac_handle* h, int T) {
    if (!h || T < 1) return AC_EINVAL;
    if (h->arch == ARCH_MIMI) return cdiv(mimi_num_frames25(h->mcfg, T), h->mcfg.resample_stride);
    if (h->arch == ARCH_DAC) return dac_num_frames(h->dcfg, T);   // 0: too short for the strided convs
    long long L = T;
    for (int r = h->cfg.num_ratios - 1; r >= 0; --r) L = (L + h->cfg.upsampling_ratios[r] - 1) / h->cfg.upsampling_ratios[r];
    return (int)L;
}
long long ac_num_samples(const ac_handle* h, int N) {
    if (!h || N < 1) return AC_EINVAL;
    return h->arch == ARCH_DAC ? dac_num_samples(h->dcfg, N) : (long long)N * h->hop;
}
int ac_hop_length(const ac_handle* h) { return h ? h->hop : AC_EINVAL; }
int ac_hidden_size(const ac_handle* h) {
    if (!h) return AC_EINVAL;
    return h->arch == ARCH_MIMI ? h->mcfg.hidden_size : h->arch == ARCH_DAC ? h->dac.H : h->cfg.hidden_size;
}
int ac_codebook_dim(const ac_handle* h) {
    if (!h) return AC_EINVAL;
    return h->arch == ARCH_MIMI ? h->mcfg.codebook_dim : h->arch == ARCH_DAC ? DAC_D : h->cfg.hidden_size;
}

static int num_q(const ac_handle* h) {
    return h->arch == ARCH_MIMI ? h->mcfg.num_quantizers : h->arch == ARCH_DAC ? h->dcfg.n_codebooks : h->cfg.num_quantizers;
}

static Workspace any_plan_ws(const ac_handle* h, int B, int T, int N, bool enc) {
    if (h->arch == ARCH_DAC) return dac_plan_ws(h, B, T, N, enc);
    if (h->arch == ARCH_WAVTOK) return wavtok_plan_ws(h, B, T, N, enc);
    return h->arch == ARCH_MIMI ? mimi_plan_ws(h, B, T, N, enc) : plan_ws(h, B, T, N, enc);
}

// DAC: clips go through the stack in chunks that keep the workspace bounded (dac_path.h)
static int dac_encode_impl(ac_handle* h, const float* sig, int B, int T, int K, float* z_out, float* zlat_out, long long* toks, float* qfeats,
                           void* ws, size_t ws_bytes, hipStream_t st) {
    const int N = dac_num_frames(h->dcfg, T);
    if (N < 1) return fail(h, AC_EINVAL, "input too short: %d samples give no frame (hop %d)", T, h->hop);
    const int Bc = dac_chunk_clips(h, B, T, 0, true), H = h->dac.H;
    for (int b0 = 0; b0 < B; b0 += Bc) {
        const int nb = std::min(Bc, B - b0);
        WsPtrs p;
        int rc = carve(h, dac_plan_ws(h, B, T, 0, true), ws, ws_bytes, &p);
        if (rc) return rc;
        float* z = z_out ? z_out + (size_t)b0 * N * H : p.act[NACT - 1];
        if (!z_out) p.used[NACT - 1] = true;
        if ((rc = amax_begin(h, st, nb))) return rc;      // every chunk starts with fresh amax slots (no activation view crosses chunks)
        rc = dac_encoder_fwd(h, st, sig + (size_t)b0 * T, nb, T, z, p);
        if (rc) return rc;
        if (zlat_out) {
            Act za{z, (long long)N * H, H, N, H};
            rc = dac_conv(h, st, h->dac.in_proj0, za, 1, 1, 1, 0, Out{zlat_out + (size_t)b0 * N * DAC_D, nullptr}, SnakeP{}, nb, nullptr);
            if (rc) return rc;
        }
        if (toks) {
            rc = dac_vq_encode(h, st, z, nb * N, K, toks + (size_t)b0 * N * K, qfeats ? qfeats + (size_t)b0 * N * H : nullptr);
            if (rc) return rc;
        }
    }
    return AC_OK;
}

size_t ac_encode_workspace_bytes(const ac_handle* h, int B, int T) {
    if (!h || B < 1 || T < 1) return 0;
    return any_plan_ws(h, B, T, 0, true).total_bytes;
}
size_t ac_decode_workspace_bytes(const ac_handle* h, int B, int N) {
    if (!h || B < 1 || N < 1) return 0;
    return any_plan_ws(h, B, 0, N, false).total_bytes;
}
size_t ac_quantizer_workspace_bytes(const ac_handle* h, int B, int N) {
    if (!h || B < 1 || N < 1 || h->arch != ARCH_MIMI) return 0;
    return align_up((size_t)B * N * 2 * h->mcfg.codebook_dim * sizeof(float) + 256, 256) + pool_bytes(1, (size_t)B * N) + 256;
}

// Mimi's quantizer entry points: scratch = [projection | split16 pool (row ring for the B*N frames)]
static float* quantizer_scratch(ac_handle* h, void* ws, int B, int N) {
    char* base = reinterpret_cast<char*>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
    const size_t proj = align_up((size_t)B * N * 2 * h->mcfg.codebook_dim * sizeof(float), 256);
    pool_bind(h, base + proj, 1, (size_t)B * N);
    return reinterpret_cast<float*>(base);
}

int ac_encode_feats(ac_handle* h, const float* sig, const float* rel_len, int B, int T, float* feats, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!sig || !feats || B < 1 || T < 1) return fail(h, AC_EINVAL, "ac_encode_feats: bad argument (B=%d, T=%d)", B, T);
    if ((rc = check_len(h, T))) return rc;
    if (!h->has_enc) return fail(h, AC_ESTATE, "ac_encode_feats: the handle was loaded without encoder weights (mode=\"decode\")");
    if (h->arch == ARCH_DAC) return dac_encode_impl(h, sig, B, T, 0, feats, nullptr, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream);
    WsPtrs p;
    rc = carve(h, any_plan_ws(h, B, T, 0, true), ws, ws_bytes, &p);
    if (rc) return rc;
    if ((rc = amax_begin(h, (hipStream_t)stream, B))) return rc;
    if (h->arch == ARCH_MIMI) return mimi_encoder_fwd(h, (hipStream_t)stream, sig, B, T, feats, p);   // no sample mask in Mimi ([HF] mimi :1245-1247)
    return encoder_fwd(h, (hipStream_t)stream, sig, rel_len, B, T, feats, p);
}

int ac_encode(ac_handle* h, const float* sig, const float* rel_len, int B, int T, int K, int64_t* toks, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!sig || !toks || B < 1 || T < 1) return fail(h, AC_EINVAL, "ac_encode: bad argument (B=%d, T=%d)", B, T);
    if (K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_encode: K=%d outside [1, %d]", K, num_q(h));
    if ((rc = check_len(h, T))) return rc;
    if (!h->has_enc) return fail(h, AC_ESTATE, "ac_encode: the handle was loaded without encoder weights (mode=\"decode\")");
    if (h->arch == ARCH_DAC)
        return dac_encode_impl(h, sig, B, T, K, nullptr, nullptr, reinterpret_cast<long long*>(toks), nullptr, ws, ws_bytes, (hipStream_t)stream);
    WsPtrs p;
    rc = carve(h, any_plan_ws(h, B, T, 0, true), ws, ws_bytes, &p);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = amax_begin(h, st, B))) return rc;
    const int N = ac_num_frames(h, T);
    if (h->arch == ARCH_MIMI) {
        const ac_mimi_config& c = h->mcfg;
        float* feats = p.act[NACT - 1];   // the encoder never has more than 5 buffers live
        p.used[NACT - 1] = true;
        rc = mimi_encoder_fwd(h, st, sig, B, T, feats, p);
        if (rc) return rc;
        float* proj = p.take();
        rc = mimi_linear(h, st, h->mimi.in_proj, feats, (long long)B * N, c.hidden_size, c.hidden_size, 0, proj, 2 * c.codebook_dim);
        if (rc) return rc;
        return mimi_rvq_encode(h, st, proj, B * N, K, reinterpret_cast<long long*>(toks));
    }
    // feats land in the activation buffer the encoder's last conv does not read from
    float* feats = p.lstm.gin;  // free again once the LSTM is done
    rc = encoder_fwd(h, st, sig, rel_len, B, T, feats, p);
    if (rc) return rc;
    return rvq_encode_fwd(h, st, feats, B * N, K, reinterpret_cast<long long*>(toks));
}

int ac_quantize_ws(ac_handle* h, const float* feats, int B, int N, int K, int64_t* toks, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!feats || !toks || B < 1 || N < 1 || K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_quantize: bad argument");
    pool_bind(h, nullptr, 0, 0);      // (EnCodec / DAC: the codebook search runs outside split-operand arithmetic)
    if (h->arch == ARCH_DAC) return dac_vq_encode(h, (hipStream_t)stream, feats, B * N, K, reinterpret_cast<long long*>(toks), nullptr);
    if (h->arch == ARCH_MIMI) {
        const ac_mimi_config& c = h->mcfg;
        if (!ws || ws_bytes < ac_quantizer_workspace_bytes(h, B, N)) return fail(h, AC_ENOMEM, "ac_quantize: workspace missing or too small");
        float* proj = quantizer_scratch(h, ws, B, N);
        if ((rc = amax_begin(h, (hipStream_t)stream, 1))) return rc;
        rc = mimi_linear(h, (hipStream_t)stream, h->mimi.in_proj, feats, (long long)B * N, c.hidden_size, c.hidden_size, 0, proj, 2 * c.codebook_dim);
        if (rc) return rc;
        return mimi_rvq_encode(h, (hipStream_t)stream, proj, B * N, K, reinterpret_cast<long long*>(toks));
    }
    return rvq_encode_fwd(h, (hipStream_t)stream, feats, B * N, K, reinterpret_cast<long long*>(toks));
}

int ac_quantize(ac_handle* h, const float* feats, int B, int N, int K, int64_t* toks, void* stream) {
    if (h && h->arch == ARCH_MIMI) return fail(h, AC_EINVAL, "ac_quantize: Mimi handles need scratch, call ac_quantize_ws");
    return ac_quantize_ws(h, feats, B, N, K, toks, nullptr, 0, stream);
}

int ac_dequantize_ws(ac_handle* h, const int64_t* toks, int B, int N, int K, float* qfeats, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!qfeats || !toks || B < 1 || N < 1 || K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_dequantize: bad argument");
    pool_bind(h, nullptr, 0, 0);
    if (h->arch == ARCH_DAC) return dac_from_codes(h, (hipStream_t)stream, reinterpret_cast<const long long*>(toks), B * N, K, qfeats);
    if (h->arch == ARCH_MIMI) {
        if (!ws || ws_bytes < ac_quantizer_workspace_bytes(h, B, N)) return fail(h, AC_ENOMEM, "ac_dequantize: workspace missing or too small");
        float* qsum = quantizer_scratch(h, ws, B, N);
        if ((rc = amax_begin(h, (hipStream_t)stream, 1))) return rc;
        return mimi_rvq_decode(h, (hipStream_t)stream, reinterpret_cast<const long long*>(toks), B * N, K, qsum, qfeats);
    }
    return rvq_decode_fwd(h, (hipStream_t)stream, reinterpret_cast<const long long*>(toks), B * N, K, qfeats);
}

int ac_dequantize(ac_handle* h, const int64_t* toks, int B, int N, int K, float* qfeats, void* stream) {
    if (h && h->arch == ARCH_MIMI) return fail(h, AC_EINVAL, "ac_dequantize: Mimi handles need scratch, call ac_dequantize_ws");
    return ac_dequantize_ws(h, toks, B, N, K, qfeats, nullptr, 0, stream);
}

int ac_decode(ac_handle* h, const int64_t* toks, int B, int N, int K, float* sig, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!sig || !toks || B < 1 || N < 1) return fail(h, AC_EINVAL, "ac_decode: bad argument (B=%d, N=%d)", B, N);
    if (K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_decode: K=%d outside [1, %d]", K, num_q(h));
    if ((rc = check_len(h, ac_num_samples(h, N)))) return rc;
    if (!h->has_dec) return fail(h, AC_ESTATE, "ac_decode: the handle was loaded without decoder weights (mode=\"encode\")");
    if (h->arch == ARCH_DAC) {
        hipStream_t st = (hipStream_t)stream;
        const int Bc = dac_chunk_clips(h, B, 0, N, false);
        const long long Lout = dac_num_samples(h->dcfg, N);
        if (Lout < 1) return fail(h, AC_EINVAL, "ac_decode: %d frames decode to no samples", N);
        for (int b0 = 0; b0 < B; b0 += Bc) {
            const int nb = std::min(Bc, B - b0);
            WsPtrs p;
            rc = carve(h, dac_plan_ws(h, B, 0, N, false), ws, ws_bytes, &p);
            if (rc) return rc;
            float* zq = p.take();
            if ((rc = amax_begin(h, st, nb))) return rc;  // every chunk starts with fresh amax slots
            rc = dac_from_codes(h, st, reinterpret_cast<const long long*>(toks) + (size_t)b0 * N * K, nb * N, K, zq);
            if (rc) return rc;
            capture(h, st, Act{zq, (long long)N * h->dac.H, h->dac.H, N, h->dac.H}, nb);
            rc = dac_decoder_fwd(h, st, zq, nb, N, sig + (size_t)b0 * Lout, p);
            if (rc) return rc;
        }
        return AC_OK;
    }
    WsPtrs p;
    rc = carve(h, any_plan_ws(h, B, 0, N, false), ws, ws_bytes, &p);
    if (rc) return rc;
    if ((rc = amax_begin(h, (hipStream_t)stream, B))) return rc;
    if (h->arch == ARCH_WAVTOK) {
        hipStream_t st = (hipStream_t)stream;
        float* zq = p.take();   // codes_to_features: the single codebook's vectors [B*N][dimension]
        rc = rvq_decode_fwd(h, st, reinterpret_cast<const long long*>(toks), B * N, K, zq);
        if (rc) return rc;
        return wavtok_decoder_fwd(h, st, zq, B, N, sig, p);
    }
    if (h->arch == ARCH_MIMI) {
        hipStream_t st = (hipStream_t)stream;
        float* qsum = p.take();
        float* qf = p.take();
        rc = mimi_rvq_decode(h, st, reinterpret_cast<const long long*>(toks), B * N, K, qsum, qf);
        if (rc) return rc;
        p.give(qsum);
        capture(h, st, Act{qf, (long long)N * h->mcfg.hidden_size, h->mcfg.hidden_size, N, h->mcfg.hidden_size}, B);
        // qf stays taken while the up-sampler reads it; the decoder needs at most 5 more buffers
        return mimi_decoder_fwd(h, st, qf, B, N, sig, p);
    }
    return decoder_fwd(h, (hipStream_t)stream, reinterpret_cast<const long long*>(toks), B, N, K, sig, p);
}

int ac_decode_feats(ac_handle* h, const float* feats, int B, int N, float* sig, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->arch != ARCH_WAVTOK) return fail(h, AC_EINVAL, "ac_decode_feats: WavTokenizer handles only (the other wrappers do not implement _feats_to_sig)");
    if (!feats || !sig || B < 1 || N < 1) return fail(h, AC_EINVAL, "ac_decode_feats: bad argument (B=%d, N=%d)", B, N);
    if ((rc = check_len(h, ac_num_samples(h, N)))) return rc;
    if (!h->has_dec) return fail(h, AC_ESTATE, "ac_decode_feats: the handle was loaded without decoder weights (mode=\"encode\")");
    WsPtrs p;
    rc = carve(h, wavtok_plan_ws(h, B, 0, N, false), ws, ws_bytes, &p);
    if (rc) return rc;
    if ((rc = amax_begin(h, (hipStream_t)stream, B))) return rc;
    return wavtok_decoder_fwd(h, (hipStream_t)stream, feats, B, N, sig, p);
}

int ac_embs(ac_handle* h, int K, float* embs, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!embs || K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_embs: bad argument");
    size_t n, off;
    if (h->arch == ARCH_MIMI) { n = (size_t)K * h->mcfg.codebook_size * h->mcfg.codebook_dim; off = h->mimi.cb_plain; }
    else if (h->arch == ARCH_DAC) { n = (size_t)K * h->dcfg.codebook_size * DAC_D; off = h->dac.cb; }
    else { n = (size_t)K * h->cfg.codebook_size * h->cfg.hidden_size; off = h->cb_plain; }
    HIPCHK(h, hipMemcpyAsync(embs, h->blob + off, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AC_OK;
}

int ac_embs_projected(ac_handle* h, int K, float* embs, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->arch == ARCH_DAC) {   // out_proj_k(codebook_k) + bias, tabulated at ac_finalize (dac.py:68-90)
        if (!embs || K < 1 || K > h->dcfg.n_codebooks) return fail(h, AC_EINVAL, "ac_embs_projected: bad argument");
        const size_t n = (size_t)K * h->dcfg.codebook_size * h->dac.H;
        HIPCHK(h, hipMemcpyAsync(embs, h->blob + h->dac.proj, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return AC_OK;
    }
    if (h->arch != ARCH_MIMI) return fail(h, AC_EINVAL, "ac_embs_projected: EnCodec has no output projection");
    const ac_mimi_config& c = h->mcfg;
    if (!embs || K < 1 || K > c.num_quantizers) return fail(h, AC_EINVAL, "ac_embs_projected: bad argument");
    // the one launching entry point without a workspace argument: its few KB of pool were allocated at ac_finalize
    pool_bind(h, h->own_pool, 1, h->own_pool_rows);
    if ((rc = amax_begin(h, (hipStream_t)stream, 1))) return rc;
    for (int q = 0; q < K; ++q) {
        const int part = q < c.num_semantic_quantizers ? 0 : 1;
        rc = mimi_linear(h, (hipStream_t)stream, h->mimi.out_proj, h->blob + h->mimi.cb_plain + (size_t)q * c.codebook_size * c.codebook_dim,
                         c.codebook_size, c.codebook_dim, c.codebook_dim, part * c.codebook_dim,
                         embs + (size_t)q * c.codebook_size * c.hidden_size, c.hidden_size);
        if (rc) return rc;
    }
    return AC_OK;
}

int ac_encode_quantized(ac_handle* h, const float* sig, int B, int T, int K, int64_t* toks, float* qfeats, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->arch != ARCH_DAC) return fail(h, AC_EINVAL, "ac_encode_quantized: DAC handles only (others: ac_encode + ac_dequantize_ws)");
    if (!sig || !toks || !qfeats || B < 1 || T < 1 || K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_encode_quantized: bad argument");
    if ((rc = check_len(h, T))) return rc;
    return dac_encode_impl(h, sig, B, T, K, nullptr, nullptr, reinterpret_cast<long long*>(toks), qfeats, ws, ws_bytes, (hipStream_t)stream);
}

int ac_encode_feats_latent(ac_handle* h, const float* sig, int B, int T, float* feats_latent, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->arch != ARCH_DAC) return fail(h, AC_EINVAL, "ac_encode_feats_latent: DAC handles only");
    if (!sig || !feats_latent || B < 1 || T < 1) return fail(h, AC_EINVAL, "ac_encode_feats_latent: bad argument");
    if ((rc = check_len(h, T))) return rc;
    return dac_encode_impl(h, sig, B, T, 0, nullptr, feats_latent, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

int ac_lstm_status(ac_handle* h) {
    if (!h) return AC_EINVAL;
    if (!h->lp_ctl) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return AC_EHIP;
    const bool usable = (h->arch == ARCH_ENCODEC || h->arch == ARCH_WAVTOK) && (h->enc_lstm.has_persist || h->dec_lstm.has_persist) && h->num_cus == 256 && !h->lstm_step_only;
    volatile unsigned* s = h->sticky;
    if (s && (s[ST_LSTM_TIMEOUT] || s[ST_LSTM_PLACEMENT]))
        return fail(h, AC_EHIP, "persistent LSTM: %u bounded waits expired, %u misplaced launches (outputs of those calls are NaN)", s[ST_LSTM_TIMEOUT], s[ST_LSTM_PLACEMENT]);
    return usable ? 1 : 0;
}

int ac_debug_split_row(const float* w, int n, uint16_t* hi, uint16_t* lo) {
    if (!w || n < 1) return AC_EINVAL;
    const int s = Packer::row_scale(w, (size_t)n);
    for (int k = 0; k < n; ++k) {
        uint16_t t[3];
        Packer::split16h(w[k], s, t);
        if (hi) hi[k] = t[0];
        if (lo) lo[k] = t[1];
    }
    return s;
}

int ac_debug_bounds(const ac_handle* h, float* out, int cap) {
    // Layout (EnCodec handles; 0 where a half was not loaded):
    //   [0..6]   enc_front.h: sb0 sb1 | hb0 hb1 | fb0 fb1h fb1x        |x0| <= sb0 + sb1 amax(sig); |h| <= hb0 + hb1 X0; |y1| <= fb0 + fb1h H + fb1x X0
    //   [7..10]  dec_tail.h:  ub0 ub1 | hb0 hb1                        |u| <= ub0 + ub1 amax(xe);   |h| <= hb0 + hb1 U
    //   [11 + 2i], [12 + 2i]  hb0, hb1 of encoder residual block i, then of decoder residual block i (AC_MAX_RATIOS each):
    //            |hidden| <= hb0 + hb1 amax(block input)  (rb_fused6.h / rb_fused6_128.h)
    if (!h || !out || !h->finalized || h->arch != ARCH_ENCODEC) return AC_EINVAL;
    const int n = 11 + 4 * AC_MAX_RATIOS;
    if (cap < n) return AC_EINVAL;
    for (int i = 0; i < n; ++i) out[i] = 0.f;
    const int last = h->cfg.num_ratios - 1;
    if (h->enc_front.ok) {
        const float v[7] = {h->enc_front.sb0, h->enc_front.sb1, h->enc_rb[0].hb0, h->enc_rb[0].hb1, h->enc_front.fb0, h->enc_front.fb1h, h->enc_front.fb1x};
        for (int i = 0; i < 7; ++i) out[i] = v[i];
    }
    if (h->dec_tail.ok) {
        const float v[4] = {h->dec_tail.sb0, h->dec_tail.sb1, h->dec_rb[last].hb0, h->dec_rb[last].hb1};
        for (int i = 0; i < 4; ++i) out[7 + i] = v[i];
    }
    for (int i = 0; i < h->cfg.num_ratios; ++i) {
        if (h->has_enc && h->enc_rb[i].has6) { out[11 + 2 * i] = h->enc_rb[i].hb0; out[12 + 2 * i] = h->enc_rb[i].hb1; }
        if (h->has_dec && h->dec_rb[i].has6) { out[11 + 2 * AC_MAX_RATIOS + 2 * i] = h->dec_rb[i].hb0; out[12 + 2 * AC_MAX_RATIOS + 2 * i] = h->dec_rb[i].hb1; }
    }
    return n;
}

int ac_debug_clock(ac_handle* h, int enable, double* shader_mhz) {
    if (!h || !h->finalized) return AC_EINVAL;
    if (shader_mhz) *shader_mhz = 0.0;
    if (h->clk_dev) {
        unsigned long long v[2] = {0, 0};
        HIPCHK(h, hipDeviceSynchronize());
        HIPCHK(h, hipMemcpy(v, h->clk_dev, sizeof v, hipMemcpyDeviceToHost));
        if (shader_mhz && v[1]) *shader_mhz = 100.0 * (double)v[0] / (double)v[1];
        HIPCHK(h, hipMemset(h->clk_dev, 0, sizeof v));
    }
    if (enable && !h->clk_dev) {
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->clk_dev), 16));
        HIPCHK(h, hipMemset(h->clk_dev, 0, 16));
    } else if (!enable && h->clk_dev) {
        (void)hipFree(h->clk_dev);
        h->clk_dev = nullptr;
    }
    return AC_OK;
}

int ac_debug_capture(ac_handle* h, float* buf_dev, size_t cap_floats) {
    if (!h) return AC_EINVAL;
    h->dbg = buf_dev;
    h->dbg_cap = buf_dev ? cap_floats : 0;
    h->dbg_used = 0;
    return AC_OK;
}

size_t ac_debug_captured(const ac_handle* h) { return h ? h->dbg_used : 0; }

int ac_profile_begin(ac_handle* h) {
    if (!h) return AC_EINVAL;
    h->prof = true;
    const char* det = std::getenv("AC_PROF_DETAIL");
    h->prof_detail = det && det[0] == '1';
    h->recs.clear();
    h->ev_used = 0;
    return AC_OK;
}

int ac_profile_end(ac_handle* h, ac_kernel_stat* out, int cap) {
    if (!h) return AC_EINVAL;
    h->prof = false;
    std::vector<ac_kernel_stat> st(h->prof_names.size());
    for (size_t i = 0; i < st.size(); ++i) {
        std::memset(&st[i], 0, sizeof st[i]);
        std::snprintf(st[i].name, sizeof st[i].name, "%s", h->prof_names[i].c_str());
    }
    for (const ProfRec& r : h->recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) return fail(h, AC_EHIP, "hipEventSynchronize failed");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return fail(h, AC_EHIP, "hipEventElapsedTime failed");
        st[r.name_id].launches += r.count;
        st[r.name_id].total_ms += ms;
        st[r.name_id].flops += r.flops;
        st[r.name_id].bytes += r.bytes;
    }
    h->recs.clear();
    h->ev_used = 0;
    int n = 0;
    for (size_t i = 0; i < st.size() && n < cap; ++i)
        if (st[i].launches) out[n++] = st[i];
    return n;
}

int ac_resample(const float* x, int B, int L, const float* kern, int n, int o, int taps, int width, float* y, int L_out, void* stream) {
    if (!x || !kern || !y || B < 1 || L < 1 || n < 1 || o < 1 || taps < 1 || width < 0 || L_out < 1) return AC_EINVAL;
    ResampleParams p{x, kern, y, B, L, L_out, n, o, taps, width};
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((L_out + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? AC_OK : AC_EHIP;
}

const char* ac_last_error(const ac_handle* h) { return h ? h->err.c_str() : "null handle"; }

void ac_destroy(ac_handle* h) {
    if (!h) return;
    if (h->blob) (void)hipFree(h->blob);
    if (h->lp_ctl) (void)hipFree(h->lp_ctl);
    if (h->own_pool) (void)hipFree(h->own_pool);
    if (h->sticky) (void)hipHostFree(h->sticky);
    if (h->clk_dev) (void)hipFree(h->clk_dev);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    delete h;
}

}  // extern "C"
