// The stream kernels of the thin stages (round 6): one wave = one stream, weights in LDS, sixteen waves per CU.
// One translation unit of the library (core.h has the map): owns the kernels of rb_stream6.h and their launchers -- kept apart from core.hip
// so that a change to these kernels recompiles in seconds.
#include "core.h"
#include "rb_stream6.h"
#include "enc_stream.h"
#include "dec_stream.h"
#include "rb_stream6m.h"
#include "rb_stream128m.h"

namespace acimpl {

// The 64-channel causal residual block (EncodecResnetBlock, [HF] modeling_encodec.py:252-282; Mimi's identity-shortcut form).  `p` arrives
// filled by launch_rb_fused6 (core.hip): tensors, amax slots, scales and bounds; here the permuted images, the segment geometry, the launch.
int launch_rb_stream6(ac_handle* h, hipStream_t st, RbFused6Params& p, const ResBlockPlan& rb, bool sc, Out out, int B) {
    constexpr int WAVES = Rs6Cfg<true>::WAVES;
    p.w3f = reinterpret_cast<const __bf16*>(h->blob + rb.w3p_off);
    p.wff = reinterpret_cast<const __bf16*>(h->blob + rb.wfp_off);
    // segments are sized so that one round of the chip's 4096 waves covers the batch (the last segment of a clip is the short one)
    const int tiles = cdiv(p.L, 16);
    const int want = std::max(1, 256 * WAVES / std::max(1, B));
    const int seg_tiles = h->dev.rb_stream > 1 ? std::min(tiles, h->dev.rb_stream - 1) : cdiv(tiles, std::min(tiles, want));   // (rb_stream = n + 1: n tiles per segment -- developer probe)
    p.seg_rows = seg_tiles * 16;
    p.nseg = cdiv(tiles, seg_tiles);
    const long long segs = (long long)B * p.nseg;
    const int grid = (int)std::min<long long>(256, (segs + WAVES - 1) / WAVES);
    const double L = p.L;
    ProfScope ps(h, st, sc ? "rb_stream6_kernel<true>" : "rb_stream6_kernel<false>",
                 2.0 * B * L * (32.0 * 192 + 64.0 * (32 + (sc ? 64 : 0))),
                 (double)B * L * 64 * 4.0 * (1 + (out.raw ? 1 : 0) + (out.elu ? 1 : 0)));
    auto go = [&](auto kern, size_t lds) -> int {
        if (int rc = ensure_lds(h, reinterpret_cast<const void*>(kern), lds)) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, st, p);
        return AC_OK;
    };
    if (sc) {
        constexpr size_t lds = Rs6Cfg<true>::lds_bytes;
        if (out.raw && out.elu) return go(rb_stream6_kernel<true, true, true>, lds);
        if (out.elu) return go(rb_stream6_kernel<true, false, true>, lds);
        return go(rb_stream6_kernel<true, true, false>, lds);
    }
    constexpr size_t lds = Rs6Cfg<false>::lds_bytes;
    if (out.raw && out.elu) return go(rb_stream6_kernel<false, true, true>, lds);
    if (out.elu) return go(rb_stream6_kernel<false, false, true>, lds);
    return go(rb_stream6_kernel<false, true, false>, lds);
}


// ---- Mimi's 64-channel identity block with its neighbour folded in (rb_stream6m.h)
int launch_rb_stream6m(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const float* xr, const float* sig, int B, int L, Out out, float* head_y, int head_k,
                       const unsigned* amax_in, unsigned* amax_out) {
    const MimiPlan::StreamM& sm = h->mimi.sm;
    const bool stem = sig != nullptr, head = head_y != nullptr;
    RbStreamMParams p{};
    p.xr = xr;
    p.sig = sig;
    p.w0f = reinterpret_cast<const __bf16*>(h->blob + sm.stem_f);
    p.w3f = reinterpret_cast<const __bf16*>(h->blob + rb.w3p_off);
    p.wff = reinterpret_cast<const __bf16*>(h->blob + rb.wfp_off);
    p.whf = reinterpret_cast<const __bf16*>(h->blob + sm.head_f);
    p.b0 = h->blob + h->mimi.enc_stem.b_off;
    p.winv0 = h->blob + sm.stem_inv;
    p.b3 = h->blob + rb.c3.b_off;
    p.winv3 = h->blob + rb.winv3_off;
    p.bf = h->blob + rb.fused.b_off;
    p.winvf = h->blob + rb.winvf_off;
    p.bh = h->blob + h->mimi.dec_head.b_off;
    p.winvh = h->blob + sm.head_inv;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.head_y = head_y;
    p.head_k = head_k;
    p.B = B;
    p.L = L;
    const int tiles = cdiv(L, 16);
    const int want = std::max(1, 256 * 16 / std::max(1, B));
    const int seg_tiles = std::max(head ? 8 : 1, cdiv(tiles, std::min(tiles, want)));     // (HEAD: a segment pays one warm-up tile)
    p.seg_rows = seg_tiles * 16;
    p.nseg = cdiv(tiles, seg_tiles);
    p.amax_in = amax_in;
    p.amax_out = amax_out;
    p.sb0 = sm.sb0; p.sb1 = sm.sb1;
    p.hb0 = rb.hb0; p.hb1 = rb.hb1;
    p.fb0 = sm.fb0; p.fb1h = sm.fb1h;
    const long long segs = (long long)B * p.nseg;
    const int grid = (int)std::min<long long>(256, (segs + 15) / 16);
    const double Ld = L;
    ProfScope ps(h, st, stem ? "rb_stream6m_kernel<stem>" : head ? "rb_stream6m_kernel<head>" : "rb_stream6m_kernel<>",
                 2.0 * B * Ld * (32.0 * 192 + 64.0 * 32 + (stem ? 64.0 * 7 : 0.0) + (head ? 64.0 * head_k : 0.0)),
                 (double)B * Ld * 4.0 * ((stem ? 1 : 64) + (head ? 1 : 64 * ((out.raw ? 1 : 0) + (out.elu ? 1 : 0)))));
    auto go = [&](auto kern) -> int {
        if (int rc = ensure_lds(h, reinterpret_cast<const void*>(kern), RM_LDS)) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), RM_LDS, st, p);
        return AC_OK;
    };
    if (head) return go(rb_stream6m_kernel<false, true, false, false>);
    if (stem) {
        if (out.raw && out.elu) return go(rb_stream6m_kernel<true, false, true, true>);
        if (out.elu) return go(rb_stream6m_kernel<true, false, false, true>);
        return go(rb_stream6m_kernel<true, false, true, false>);
    }
    return fail(h, AC_ESTATE, "rb_stream6m without a folded layer: use rb_stream6");
}

// ---- the 128-channel residual block without a slab (rb_stream128m.h): Mimi's identity form at sixteen waves per CU, EnCodec's 1x1-shortcut
// form (its shortcut weights read from L2 per tile) at twelve
template <int WAVES, bool SC>
static int rb_stream128m_go(ac_handle* h, hipStream_t st, RbStream128Params& p, Out out, int B) {
    const int tiles = cdiv(p.L, 16);
    const int want = std::max(1, 256 * WAVES / std::max(1, B));
    const int seg_tiles = cdiv(tiles, std::min(tiles, want));
    p.seg_rows = seg_tiles * 16;
    p.nseg = cdiv(tiles, seg_tiles);
    const long long segs = (long long)B * p.nseg;
    const int grid = (int)std::min<long long>(256, (segs + WAVES - 1) / WAVES);
    const double L = p.L;
    ProfScope ps(h, st, SC ? "rb_stream128m_kernel<true>" : "rb_stream128m_kernel<false>", 2.0 * B * L * (64.0 * 384 + 128.0 * (64 + (SC ? 128 : 0))),
                 (double)B * L * 128 * 4.0 * (1 + (out.raw ? 1 : 0) + (out.elu ? 1 : 0)));
    auto go = [&](auto kern) -> int {
        if (int rc = ensure_lds(h, reinterpret_cast<const void*>(kern), r128_lds<WAVES>())) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), r128_lds<WAVES>(), st, p);
        return AC_OK;
    };
    if (out.raw && out.elu) return go(rb_stream128m_kernel<WAVES, SC, true, true>);
    if (out.elu) return go(rb_stream128m_kernel<WAVES, SC, false, true>);
    return go(rb_stream128m_kernel<WAVES, SC, true, false>);
}
int launch_rb_stream128m(ac_handle* h, hipStream_t st, RbFused6Params& q, const ResBlockPlan& rb, bool sc, Out out, int B) {
    RbStream128Params p{};
    p.xr = q.xr;
    p.w3f = reinterpret_cast<const __bf16*>(h->blob + rb.w3p_off);
    p.wff = reinterpret_cast<const __bf16*>(h->blob + rb.wfp_off);
    p.b3 = q.b3; p.winv3 = q.winv3; p.bf = q.bf; p.winvf = q.winvf;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.B = B;
    p.L = q.L;
    p.amax_in = q.amax_in;
    p.amax_out = q.amax_out;
    p.hb0 = q.hb0; p.hb1 = q.hb1;
    p.pad = q.pad; p.Lp = q.Lp;
    if (sc) return rb_stream128m_go<12, true>(h, st, p, out, B);     // (the two fragment sets of the shortcut do not fit 128 registers)
    return rb_stream128m_go<16, false>(h, st, p, out, B);
}

// ---- the encoder's thin-channel head (enc_stream.h): stem -> ResBlock(32) -> ELU -> Conv1d(32, 64, k4, s2); called by enc_front_fwd (core.hip),
// which owns the shape checks, the amax slots and the output descriptors
int enc_stream_fwd(ac_handle* h, hipStream_t st, const float* sig, const float* rel_len, int B, int T, float* y, float* dbg_x0, float* dbg_y1,
                   const unsigned* amax_sig, unsigned* amax_out) {
    const ResBlockPlan& rb = h->enc_rb[0];
    const PackedGemm& gd = h->enc_down[0];
    EncStreamParams p{};
    p.sig = sig;
    p.rel_len = rel_len;
    p.w0f = reinterpret_cast<const __bf16*>(h->blob + h->simg.stem_f);
    p.w3f = reinterpret_cast<const __bf16*>(h->blob + rb.w3p_off);
    p.wff = reinterpret_cast<const __bf16*>(h->blob + rb.wfp_off);
    p.wdf = reinterpret_cast<const __bf16*>(h->blob + h->simg.down_f);
    p.b0 = h->blob + h->enc_stem.b_off;
    p.winv0 = h->blob + h->simg.stem_inv;
    p.b3 = h->blob + rb.c3.b_off;
    p.winv3 = h->blob + rb.winv3_off;
    p.bf = h->blob + rb.fused.b_off;
    p.winvf = h->blob + rb.winvf_off;
    p.bd = h->blob + gd.b_off;
    p.winvd = h->blob + h->simg.down_inv;
    p.y = y;
    p.dbg_x0 = dbg_x0;
    p.dbg_y1 = dbg_y1;
    p.B = B;
    p.T = T;
    p.M = cdiv(T, 2);
    const int nchunks = cdiv(T, 32);
    p.seg_chunks = std::max(8, cdiv(nchunks, std::max(1, 256 * ES_WAVES / B)));     // one round of the chip's 4096 waves
    if (h->dev.front_seg > 0) p.seg_chunks = h->dev.front_seg;
    p.segs_per_clip = cdiv(nchunks, p.seg_chunks);
    p.amax_sig = amax_sig;
    p.amax_out = amax_out;
    p.sb0 = h->enc_front.sb0; p.sb1 = h->enc_front.sb1;
    p.hb0 = rb.hb0; p.hb1 = rb.hb1;
    p.fb0 = h->enc_front.fb0; p.fb1h = h->enc_front.fb1h; p.fb1x = h->enc_front.fb1x;
    if (int rc = ensure_lds(h, reinterpret_cast<const void*>(enc_stream_kernel), ES_LDS)) return rc;
    const long long streams = (long long)B * p.segs_per_clip;
    ProfScope ps(h, st, "enc_stream_kernel", 2.0 * B * (double)T * (7.0 * 32 + 16.0 * 96 + 32.0 * 48 + 64.0 * 128 / 2),
                 (double)B * T * 4.0 + (double)B * p.M * 256.0);
    hipLaunchKernelGGL(enc_stream_kernel, dim3((unsigned)cdiv((int)streams, ES_WAVES)), dim3(64 * ES_WAVES), ES_LDS, st, p);
    return AC_OK;
}


// ---- the decoder's thin-channel tail (dec_stream.h): ConvTranspose1d(64, 32, k4, s2) -> ResBlock(32) -> ELU -> Conv1d(32, 1, k7); called by
// dec_tail_fwd (core.hip)
int dec_stream_fwd(ac_handle* h, hipStream_t st, const Act& xe, int B, float* sig, float* dbg_u, float* dbg_v, const unsigned* amax_x) {
    const int last = h->cfg.num_ratios - 1;
    const ResBlockPlan& rb = h->dec_rb[last];
    const PackedGemm& gu = h->dec_up[last];
    DecStreamParams p{};
    p.xe = xe.p;
    p.wuf = reinterpret_cast<const __bf16*>(h->blob + h->simg.up_f);
    p.w3f = reinterpret_cast<const __bf16*>(h->blob + rb.w3p_off);
    p.wff = reinterpret_cast<const __bf16*>(h->blob + rb.wfp_off);
    p.whf = reinterpret_cast<const __bf16*>(h->blob + h->simg.head_f);
    p.bu = h->blob + gu.b_off;
    p.winvu = h->blob + h->simg.up_inv;
    p.b3 = h->blob + rb.c3.b_off;
    p.winv3 = h->blob + rb.winv3_off;
    p.bf = h->blob + rb.fused.b_off;
    p.winvf = h->blob + rb.winvf_off;
    p.bh = h->blob + h->dec_head.b_off;
    p.winvh = h->blob + h->simg.head_inv;
    p.sig = sig;
    p.dbg_u = dbg_u;
    p.dbg_v = dbg_v;
    p.B = B;
    p.L = xe.L;
    const int nchunks = cdiv(xe.L, 16);
    p.seg_chunks = std::max(8, cdiv(nchunks, std::max(1, 256 * DS_WAVES / B)));
    if (h->dev.tail_seg > 0) p.seg_chunks = h->dev.tail_seg;
    p.segs_per_clip = cdiv(nchunks, p.seg_chunks);
    p.amax_x = amax_x;
    p.ub0 = h->dec_tail.sb0; p.ub1 = h->dec_tail.sb1;
    p.hb0 = rb.hb0; p.hb1 = rb.hb1;
    p.fb0 = h->dec_tail.fb0; p.fb1h = h->dec_tail.fb1h; p.fb1x = h->dec_tail.fb1x;
    if (int rc = ensure_lds(h, reinterpret_cast<const void*>(dec_stream_kernel), DS_LDS)) return rc;
    const long long streams = (long long)B * p.segs_per_clip;
    ProfScope ps(h, st, "dec_stream_kernel", 2.0 * B * (double)xe.L * (64.0 * 128 + 2.0 * (16.0 * 96 + 32.0 * 48 + 7.0 * 32)),
                 (double)B * xe.L * 256.0 + (double)B * xe.L * 8.0);
    hipLaunchKernelGGL(dec_stream_kernel, dim3((unsigned)cdiv((int)streams, DS_WAVES)), dim3(64 * DS_WAVES), DS_LDS, st, p);
    return AC_OK;
}

}  // namespace acimpl
