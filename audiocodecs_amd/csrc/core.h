// Host-side internals shared by the translation units of libaudiocodecs_amd.so:
//   core.hip        the shared machinery (weight packing, split16 pools, tap-GEMM dispatch, fused-block / LSTM / codebook launchers,
//                   workspace planning) and the EnCodec encoder / decoder; owns every kernel the codecs share
//   mimi_path.hip, dac_path.hip, wavtok_path.hip   one codec each: plan, finalize, forward passes, its own kernels, its ac_*_create
//   ac_api.hip      the extern "C" entry points of include/audiocodecs_amd.h
// This header declares; it includes no header that defines a non-template kernel.
#pragma once
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
#include "lstm_consts.h"
#include "rvq_types.h"
#include "split16.h"
#include "tap_gemm.h"

using namespace ac;

namespace ac {
constexpr int THIN_MAXK = 8;   // widest kernel of the dedicated stem / head kernels (thin.h)
struct RbFused6Params;         // rb_fused6.h
}

namespace acimpl {


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
    size_t w3p_off = 0, wfp_off = 0;       // rb_stream6.h (C = 64): the same images with every k-step's 32 columns in the lanes' load order
    float hb0 = 0.f, hb1 = 0.f;            // split16.h: |hidden| <= hb0 + hb1 * amax(x)
    bool has6 = false;
};

struct LstmPlan {
    int D, layers;
    std::vector<PackedGemm> ih;     // [4D][D] + (b_ih + b_hh)
    std::vector<size_t> hh_off;     // W_hh per layer, MFMA B-fragment order
    std::vector<size_t> ihpk_off;   // W_ih per layer, same order (used by the in-step projection of layers >= 1)
    size_t persist_off = 0;         // register images of W_hh0, W_ih1, W_hh1 for lstm_persist_kernel (D = 512, 2 layers)
    size_t persist16_off = 0;       // lstm_persist16_kernel's register images: two fp16 planes of the scaled rows (float offset into the blob)
    size_t persist16_inv = 0;       // [2 layers][4D] 2^-s of those rows (0: exact-product arithmetic, no images)
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
    size_t cb16 = 0, cb16_inv = 0;                     // rvq16.h: split16 images of the tables + their 2^-s (0: exact-product arithmetic)
    size_t rope_cos = 0, rope_sin = 0;                 // [rope_T][head_dim]
    int rope_T = 0;
    int D = 0;                                         // SEANet width at the bottleneck
    // rb_stream6m.h (round 6): the encoder stem as a [64][7 -> 32] operand image and the decoder head as a [k -> 16 taps][64] one (float
    // offsets into the blob, their 2^-s), the stem's bound |x0| <= sb0 + sb1 amax(sig), the last block's |y - x| <= fb0 + fb1h bound(hidden)
    struct StreamM { size_t stem_f = 0, stem_inv = 0, head_f = 0, head_inv = 0; float sb0 = 0.f, sb1 = 0.f, fb0 = 0.f, fb1h = 0.f; bool stem_ok = false, head_ok = false; } sm;
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


}  // namespace acimpl
using namespace acimpl;

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
    // enc_stream.h / dec_stream.h (round 6): split16 fragment images with every k-step's 32 columns in the lanes' load order (float offsets
    // into the blob) and their per-row 2^-s: the stem as a [32][7 -> 32] matrix, the strided conv, the transposed conv, the head as a
    // [7 -> 16 taps][32] matrix; the 32-channel blocks' images are ResBlockPlan::w3p_off / wfp_off
    struct StreamImgs { size_t stem_f = 0, stem_inv = 0, down_f = 0, down_inv = 0, up_f = 0, up_inv = 0, head_f = 0, head_inv = 0; bool enc_ok = false, dec_ok = false; } simg;
    // Developer / test switches.  Latched from the environment ONCE, at ac_finalize (latch_dev_switches, ac_api.hip); afterwards only
    // ac_debug_set changes them -- no compute entry point reads the environment (round-3 advisor finding: hundreds of getenv calls per
    // step, racing with a test's setenv, and a stray variable in a user's shell silently changing which kernel runs mid-process).
    struct DevSwitches {
        int tap_epi_staged = 0;     // AC_TAP_EPI=staged   : every tap-GEMM through the LDS-staged epilogue (A/B against the direct one)
        int tap_dil = 1;            // AC_TAP_DIL=0        : dilated taps reload the slab per tap instead of the wide-slab instantiation
        int tap_stagger = 0;        // AC_TAP_STAGGER=n    : random start delays (timing experiment)
        int tap_pick = -1;          // AC_TAP_PICK=0|1|2   : force a tile arrangement for N % 256 == 0 layers
        int tap8 = -1;              // AC_TAP8=-1|0|1      : tap_gemm8.h by the cost model (default) / never / wherever the shape allows
        int tap8_form = 0;          // AC_TAP8_FORM=1|2|3  : force its tile form (256 x 256, 256 x 128, 128 x 256) where the shape allows
        int tap8_spread = 1;        // AC_TAP8_SPREAD=0|1|2: tap_gemm8's requests of a stage at its top / dealt between its MFMA units where that measured faster (128-row tiles) / dealt everywhere (bit-identical)
        int rb6_dbg = 0;            // AC_RB6_DBG          : timing variants of the fused blocks (wrong results)
        int rb_stream = 1;          // AC_RB_STREAM=0      : the 64-channel causal blocks through rb_fused6.h instead of rb_stream6.h (A/B, cross-check tests)
        int rb128_stream = 1;       // AC_RB128_STREAM=0   : Mimi's 128-channel identity blocks through rb128_fused6 instead of rb_stream128m.h
        int chain_stream = 1;       // AC_CHAIN_STREAM=0   : the fused thin-channel chains through enc_front.h / dec_tail.h instead of enc_stream.h / dec_stream.h
        int front_seg = 0, tail_seg = 0;   // AC_FRONT_SEG / AC_TAIL_SEG: chunks per stream of the fused chains (0: from the batch size)
        int front_ldspad = 0;       // AC_FRONT_LDSPAD     : extra dynamic LDS (forces one workgroup per CU)
        int lstm_dbg = 0;           // AC_LSTM_DBG         : fault injection / traces of the persistent LSTM
        int lstm_fuse_in = 1;       // AC_LSTM_FUSE_IN=0   : layer-0 input projection as a separate GEMM
        int rvq_exact = 0;          // AC_RVQ=fp32         : exact-product codebook search
        int prof_detail = 0;        // AC_PROF_DETAIL=1    : one profile record per tap-GEMM shape
        int attn_exact = 0;         // AC_ATTN_EXACT=1     : Mimi attention on the fp32 MFMA (attention_kernel) instead of attention16_kernel
        int dac_unit = 1;           // AC_DAC_UNIT=0       : DAC's 96-channel residual units as two tap-GEMM launches instead of dac_unit6_kernel
        int mimi_tail = 1;          // AC_MIMI_TAIL=0      : Mimi's last residual block and the final conv as two kernels (rb_fused6<64,false> + head4)
        int head_seq = 0;           // AC_HEAD_SEQ=1       : the one-thread-per-sample head kernel (A/B against head4_kernel)
    } dev;
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
                                    // (false: split16.h -- fp32-fidelity arithmetic of the matrix kernels, two fp16 planes, 3 products)
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

namespace acimpl {

int fail(ac_handle* h, int code, const char* fmt, ...);


#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return fail(h, AC_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)


inline int cdiv(int a, int b) { return (a + b - 1) / b; }

int ensure_lds(ac_handle* h, const void* func, size_t bytes);

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

Arch make_arch(const ac_config& c);


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
    bool use16() const { return !h->gemm_fp32; }
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
        if (use16()) pack16(g);
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
        return 0;   // exact-product arithmetic: no fragment images
    }
    // rvq16.h: split16 image of Q codebook tables [C][H] at blob offset `plain`, one power-of-two scale per table, halves in the lanes'
    // own dim order:  [q][code tile 16][k-step s of 32][plane 2][lane (j, kq)][e 8]  <->  dim 16 (2s + e/4) + 4 kq + e%4 of code 16 ct + j
    void pack_cb16(size_t plain, int Q, int C, int H, size_t* img_off, size_t* inv_off) {
        const int KS = H / 32;
        *img_off = reserve((size_t)Q * C * H);                  // 2 planes x 2 bytes = 4 bytes per element
        *inv_off = reserve((size_t)Q);
        std::vector<uint16_t> img((size_t)Q * C * H * 2);
        for (int q = 0; q < Q; ++q) {
            const float* e = &blob[plain + (size_t)q * C * H];
            const int se = row_scale(e, (size_t)C * H);
            blob[*inv_off + q] = s16_pow2(-se);
            for (int ct = 0; ct < C / 16; ++ct)
                for (int s = 0; s < KS; ++s)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e8 = 0; e8 < 8; ++e8) {
                            const int j = lane & 15, kq = lane >> 4;
                            const int dim = 16 * (2 * s + e8 / 4) + 4 * kq + e8 % 4;
                            uint16_t t[3];
                            split16h(e[(size_t)(ct * 16 + j) * H + dim], se, t);
                            const size_t base = ((((size_t)q * (C / 16) + ct) * KS + s) * 2) * 512 + (size_t)lane * 8 + e8;
                            img[base] = t[0];
                            img[base + 512] = t[1];
                        }
        }
        std::memcpy(&blob[*img_off], img.data(), img.size() * 2);
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
        if ((C == 64 || C == 32 || C == 128) && use16()) {   // rb_stream6.h / enc_stream.h / dec_stream.h / rb_stream128m.h (identity shortcut): column 8 kq + e of a k-step <-> channel 4 kq + e (e < 4), 16 + 4 kq + e - 4 (e >= 4) of its 32
            std::vector<int> k3p(k3.size()), kfp(kf.size());
            for (size_t k = 0; k < k3.size(); ++k) k3p[k] = k3[(k & ~(size_t)31) + ((k & 4) ? 16 : 0) + 4 * ((k >> 3) & 3) + (k & 3)];
            for (size_t k = 0; k < kf.size(); ++k) kfp[k] = kf[(k & ~(size_t)31) + ((k & 4) ? 16 : 0) + 4 * ((k >> 3) & 3) + (k & 3)];
            rb.w3p_off = frag16(rb.c3.w_off, hid, rb.c3.Ktot, k3p);
            rb.wfp_off = frag16(rb.fused.w_off, C, rb.fused.Ktot, kfp);
        }
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
    // the lanes' load order of the stream kernels (rb_stream6.h): column 8 kq + e of a k-step <-> 4 kq + e (e < 4), 16 + 4 kq + e - 4 (e >= 4)
    static std::vector<int> perm32(const std::vector<int>& km) {
        std::vector<int> o(km.size());
        for (size_t k = 0; k < km.size(); ++k) o[k] = km[(k & ~(size_t)31) + ((k & 4) ? 16 : 0) + 4 * ((k >> 3) & 3) + (k & 3)];
        return o;
    }
    // rb_stream6m.h: Mimi's stem [64][7] as an MFMA operand image + its bound; the head's [k][64] taps as the rows of a [16][64] matrix + the
    // bound of the last block's residual branch
    void stream_mimi(const PackedGemm& stem, const ResBlockPlan& enc_rb, const ResBlockPlan& dec_rb, const PackedGemm& head, int head_k, MimiPlan::StreamM& sm) {
        sm.stem_ok = sm.head_ok = false;
        if (!use16()) return;
        if (stem.N == 64 && stem.Ktot == 7 && stem.has_bias && enc_rb.C == 64 && enc_rb.w3p_off) {
            std::vector<int> k7(32, -1);
            for (int k = 0; k < 7; ++k) k7[k] = k;
            sm.stem_f = frag16(stem.w_off, 64, 7, k7, &sm.stem_inv);
            sm.sb0 = sm.sb1 = 0.f;
            for (int n = 0; n < 64; ++n) {
                double l1 = 0.0;
                for (int k = 0; k < 7; ++k) l1 += std::fabs((double)blob[stem.w_off + (size_t)n * 7 + k]);
                sm.sb1 = std::max(sm.sb1, (float)(l1 * 1.000001));
                sm.sb0 = std::max(sm.sb0, std::fabs(blob[stem.b_off + n]));
            }
            sm.stem_ok = true;
        }
        if (head.N == 1 && head_k >= 1 && head_k <= 8 && head.Ktot == head_k * 64 && head.has_bias && dec_rb.C == 64 && dec_rb.w3p_off && dec_rb.fused.Ktot == 32) {
            std::vector<int> kh(64);
            for (int k = 0; k < 64; ++k) kh[k] = k;
            const size_t pad = reserve(16 * 64);
            for (int i = 0; i < 16 * 64; ++i) blob[pad + i] = i < head_k * 64 ? blob[head.w_off + i] : 0.f;
            sm.head_f = frag16(pad, 16, 64, perm32(kh), &sm.head_inv);
            sm.fb0 = sm.fb1h = 0.f;
            for (int n = 0; n < 64; ++n) {
                double lh = 0.0;
                for (int k = 0; k < 32; ++k) lh += std::fabs((double)blob[dec_rb.fused.w_off + (size_t)n * 32 + k]);
                sm.fb1h = std::max(sm.fb1h, (float)(lh * 1.000001));
                sm.fb0 = std::max(sm.fb0, std::fabs(blob[dec_rb.fused.b_off + n]));
            }
            sm.head_ok = true;
        }
    }
    // enc_stream.h: the stem as a [32][7 -> 32] MFMA operand (tap j in column j), the strided conv [64][128] permuted
    void stream_enc(const PackedGemm& stem, const ResBlockPlan& rb, const PackedGemm& down, ac_handle::StreamImgs& si) {
        si.enc_ok = false;
        if (!use16() || !h->enc_front.ok || !rb.w3p_off || stem.N != 32 || stem.Ktot != 7 || down.N != 64 || down.Ktot != 128) return;
        std::vector<int> k7(32, -1), kd(128);
        for (int k = 0; k < 7; ++k) k7[k] = k;
        for (int k = 0; k < 128; ++k) kd[k] = k;
        si.stem_f = frag16(stem.w_off, 32, 7, k7, &si.stem_inv);
        si.down_f = frag16(down.w_off, 64, 128, perm32(kd), &si.down_inv);
        si.enc_ok = true;
    }
    // dec_stream.h: the transposed conv [64][128] permuted, the head's [7][32] taps as the rows of a [16][32] matrix (rows 7..15 zero)
    void stream_dec(const PackedGemm& up, const ResBlockPlan& rb, const PackedGemm& head, ac_handle::StreamImgs& si) {
        si.dec_ok = false;
        if (!use16() || !h->dec_tail.ok || !rb.w3p_off || up.N != 64 || up.Ktot != 128 || head.N != 1 || head.Ktot != 7 * 32) return;
        std::vector<int> ku(128), kh(32);
        for (int k = 0; k < 128; ++k) ku[k] = k;
        for (int k = 0; k < 32; ++k) kh[k] = k;
        si.up_f = frag16(up.w_off, 64, 128, perm32(ku), &si.up_inv);
        const size_t pad = reserve(16 * 32);
        for (int i = 0; i < 16 * 32; ++i) blob[pad + i] = i < 7 * 32 ? blob[head.w_off + i] : 0.f;
        si.head_f = frag16(pad, 16, 32, perm32(kh), &si.head_inv);
        si.dec_ok = true;
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
        // dec_stream.h: |block out| <= fb0 + fb1h bound(hidden) + fb1x bound(block in) -- the scale of ELU(v) as the head's MFMA operand
        const int hid = rb.C / 2;
        cb.fb0 = cb.fb1h = cb.fb1x = 0.f;
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
            if (use16()) {
                // lstm_persist16_kernel: [hh0, ih1, hh1, ih0][32 slices][4 waves][4 gates][4 k-steps of 32][2 planes][64 lanes][8 fp16] of the
                // scaled rows; the two matrices of a layer share the accumulator, so their rows share the scale
                const size_t mat16 = (size_t)LP_SLICES * 4 * 4 * 4 * 2 * 512;
                lp.persist16_off = reserve((4 * mat16 + 1) / 2);
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
                std::memcpy(&blob[lp.persist16_off], pl16.data(), pl16.size() * 2);
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

int prof_name(ac_handle* h, const char* nm);

hipEvent_t next_event(ac_handle* h);


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

void pool_bind(ac_handle* h, void* mem, int pool_B, size_t rows);

int amax_begin(ac_handle* h, hipStream_t st, int B);

unsigned* amax_new(ac_handle* h);

const unsigned* amax_of(ac_handle* h, hipStream_t st, const float* x, long long bs, long long ts, int L, int C, int B, const unsigned* known);

unsigned* rowmax_new(ac_handle* h, hipStream_t st, long long rows, bool zero);

const unsigned* amax_plus(ac_handle* h, hipStream_t st, const Act& x, float add, int B);

TapSeg make_seg(const Act& x, int s, int J, int pad /*PAD_**/, int extra, int kofs, const float* rel_len, int left = -1, int right = 0);


inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int run_tap(ac_handle* h, hipStream_t st, TapGemmParams& p);


// What a layer should produce: the raw output, its ELU, or both (SEANet consumers all start with ELU;
// shortcuts, LSTMs and the public outputs want the raw value).
struct Out {
    float* raw = nullptr;
    float* elu = nullptr;
};

struct Act2 {       // a layer output in up to two flavours (same shape/strides)
    Act raw{}, elu{};
};

int try_thin6(ac_handle* h, hipStream_t st, const PackedGemm& g, const Act& x, int width, int edge, Out out, int B, const unsigned** amax_out = nullptr);

int conv_fwd(ac_handle* h, hipStream_t st, const PackedGemm& g, const Act& x, int k, int s, const float* rel_len, Out out,
             long long out_bs, long long out_rs, int B, Act2* y);

int convtr_fwd(ac_handle* h, hipStream_t st, const PackedGemm& g, const Act& x, int s, Out out, int B, Act2* y);

bool rb128_ok(const ac_handle* h, const ResBlockPlan& rb);

int resblock_fwd(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, float* hbuf, Out out, int B, Act2* y);

bool thin_ok(const ac_config& c, int k);

int thin_stem(ac_handle* h, hipStream_t st, const PackedGemm& g, int F, int k, int pad, const float* sig, const float* rel_len, int B, int T,
              Out out, Act2* y, int padl = -1, const float* alpha = nullptr, const float* alpha_inv = nullptr, int Lp = -1);

int stem_fwd(ac_handle* h, hipStream_t st, const float* sig, const float* rel_len, int B, int T, Out out, Act2* y);

int thin_head(ac_handle* h, hipStream_t st, const PackedGemm& g, int F, int k, int pad, const Act& x, int B, float* sig, int padl = -1,
              int tanh_out = 0);

int head_fwd(ac_handle* h, hipStream_t st, const Act& x, int B, float* sig);

bool enc_front_ok(const ac_handle* h, int T);

int enc_front_fwd(ac_handle* h, hipStream_t st, const float* sig, const float* rel_len, int B, int T, float* y, float* dbg_x0, float* dbg_y1, Act2* out);

bool dec_tail_ok(const ac_handle* h, const Act& xe);

int dec_tail_fwd(ac_handle* h, hipStream_t st, const Act& xe, int B, float* sig, float* dbg_u, float* dbg_v);

void capture(ac_handle* h, hipStream_t st, const Act& a, int B);


struct LstmWs {
    float *gin, *gin1, *hseq0, *hseq1, *c;   // c holds one [B][D] cell state per layer
};

int lstm_fwd(ac_handle* h, hipStream_t st, const LstmPlan& lp, const Act& x, const LstmWs& ws, Out out, int B, Act2* y);

int rvq_encode_fwd(ac_handle* h, hipStream_t st, const float* feats, int F, int K, long long* toks);

int rvq_decode_fwd(ac_handle* h, hipStream_t st, const long long* toks, int F, int K, float* out);


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

Workspace plan_ws(const ac_handle* h, int B, int T_in /*samples, encoder*/, int N_frames /*decoder*/, bool enc);


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

int carve(ac_handle* h, const Workspace& w, void* ws, size_t ws_bytes, WsPtrs* o);

int check_ready(ac_handle* h);

int check_len(ac_handle* h, long long samples);

int encoder_fwd(ac_handle* h, hipStream_t st, const float* sig, const float* rel_len, int B, int T, float* feats, WsPtrs& ws);

int decoder_fwd(ac_handle* h, hipStream_t st, const long long* toks, int B, int N, int K, float* sig, WsPtrs& ws);

// ---- wrappers around launches whose kernels live in core.hip (the per-codec translation units never include a header that
// DEFINES a non-template kernel: one definition per library)
bool rb64_identity_head_fwd(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, const PackedGemm& head, int head_k, float* sig, int B, int* rc);
int enc_stream_fwd(ac_handle* h, hipStream_t st, const float* sig, const float* rel_len, int B, int T, float* y, float* dbg_x0, float* dbg_y1, const unsigned* amax_sig, unsigned* amax_out);   // stream_path.hip
int dec_stream_fwd(ac_handle* h, hipStream_t st, const Act& xe, int B, float* sig, float* dbg_u, float* dbg_v, const unsigned* amax_x);   // stream_path.hip
// rb_stream6m.h (stream_path.hip): Mimi's 64-channel identity block with the stem (sig != null: x is computed from the samples) or the head
// (head_y != null: one sample per row is stored instead of the block's output) folded in
int launch_rb_stream6m(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const float* xr, const float* sig, int B, int L, Out out, float* head_y, int head_k,
                       const unsigned* amax_in, unsigned* amax_out);
// rb_stream128m.h (stream_path.hip): the 128-channel block without a slab (Mimi's identity form, EnCodec's 1x1-shortcut form); `p` arrives
// filled by launch_rb128_fused6 (core.hip)
int launch_rb_stream128m(ac_handle* h, hipStream_t st, RbFused6Params& p, const ResBlockPlan& rb, bool sc, Out out, int B);
int launch_rb_stream6(ac_handle* h, hipStream_t st, RbFused6Params& p, const ResBlockPlan& rb, bool sc, Out out, int B);   // stream_path.hip (p filled by launch_rb_fused6)
int rb64_identity_fwd(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, Out out, int B, const unsigned** amax_out);   // rb_fused6<64, false> / rb_fused<64,64,2,false>
int rb128_identity_fwd(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, Out out, int B, const unsigned** amax_out); // rb128_fused6<false>
int rvq_encode_cdist_launch(ac_handle* h, hipStream_t st, const RvqEncParams& p, unsigned blocks, const _Float16* epk16 = nullptr, const float* einv = nullptr);   // rvq_encode16_kernel<16, 1, true> (images given, H = 256) or rvq_encode_kernel<H/16, 1, true>
void rvq_decode_launch(hipStream_t st, const RvqDecParams& p, unsigned blocks);
void amax_fill_launch(hipStream_t st, unsigned* slot, unsigned bits, int B);
int resample_launch(const float* x, int B, int L, const float* kern, int n, int o, int taps, int width, float* y, int L_out, hipStream_t st);

// ---- per-codec paths (mimi_path.hip, dac_path.hip, wavtok_path.hip)
// optional epilogue terms of a tap-GEMM launch (Mimi's transformers, WavTokenizer's backbone, DAC's residual units)
struct Epi {
    const float* scale = nullptr;
    const float* res = nullptr;
    long long res_bs = 0, res_rs = 0;
    int gelu = 0;
    // split16.h row mode (mimi_linear): per-row amax words of the input when the producing linear layer left them, and
    // where to return the output's (null: not wanted)
    const unsigned* rowmax_in = nullptr;
    const unsigned** rowmax_out = nullptr;
};

int mimi_finalize(ac_handle* h, Packer& pk);
Workspace mimi_plan_ws(const ac_handle* h, int B, int T_in, int N_frames, bool enc);
int mimi_num_frames25(const ac_mimi_config& c, long long T);
int mimi_encoder_fwd(ac_handle* h, hipStream_t st, const float* sig, int B, int T, float* feats, WsPtrs& ws);
int mimi_decoder_fwd(ac_handle* h, hipStream_t st, const float* qfeats, int B, int N, float* sig, WsPtrs& ws);
// y[rows][N] = epi(x[rows][cin-slice] * W^T): a 1-tap GEMM over a merged token matrix (core.hip; Mimi and WavTokenizer use it)
int mimi_linear(ac_handle* h, hipStream_t st, const PackedGemm& g, const float* x, long long rows, int cin, int x_pitch, int kofs,
                float* y, int y_pitch, const Epi& epi = Epi{});
int layernorm_fwd(ac_handle* h, hipStream_t st, const float* x, size_t w_off, size_t b_off, float* y, long long rows, int H, float eps,
                  const unsigned** rows_out = nullptr);   // mimi_path.hip
int mimi_rvq_encode(ac_handle* h, hipStream_t st, const float* proj, int F, int K, long long* toks);
int mimi_rvq_decode(ac_handle* h, hipStream_t st, const long long* toks, int F, int K, float* qsum, float* qfeats);
int dac_finalize(ac_handle* h, Packer& pk);
Workspace dac_plan_ws(const ac_handle* h, int B, long long T_in, long long N_frames, bool enc);
int dac_chunk_clips(const ac_handle* h, int B, long long T_in, long long N_frames, bool enc);
int dac_num_frames(const ac_dac_config& c, long long T);
long long dac_num_samples(const ac_dac_config& c, long long N);
int dac_encoder_fwd(ac_handle* h, hipStream_t st, const float* sig, int B, int T, float* z, WsPtrs& ws);
int dac_decoder_fwd(ac_handle* h, hipStream_t st, const float* zq, int B, int N, float* sig, WsPtrs& ws);
int dac_latent_proj(ac_handle* h, hipStream_t st, const float* z, int N, int nb, float* zlat);     // in_proj of the first codebook (dac.py:103-112)
int dac_vq_encode(ac_handle* h, hipStream_t st, const float* z, int F, int K, long long* toks, float* qsum);
int dac_from_codes(ac_handle* h, hipStream_t st, const long long* toks, int F, int K, float* zq);
int wavtok_finalize(ac_handle* h, Packer& pk);
Workspace wavtok_plan_ws(const ac_handle* h, int B, int T_in, int N_frames, bool enc);
int wavtok_decoder_fwd(ac_handle* h, hipStream_t st, const float* feats, int B, int N, float* sig, WsPtrs& ws);
int upload_blob(ac_handle* h, Packer& pk, int device);
constexpr int DAC_CODE_DIM = 8;   // = DAC_D (dac.h), asserted in dac_path.hip

}  // namespace acimpl
