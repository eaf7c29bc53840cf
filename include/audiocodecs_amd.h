/*
 * audiocodecs_amd.h -- C ABI of the MI355X (gfx950) EnCodec encode/decode path.
 *
 * Drop-in boundary: the reference is pure Python and has no FFI of its own; the calls this
 * library replaces are the third-party model calls inside the reference wrapper
 *     audiocodecs/encodec.py:90-93   self.model.encode(sig[:, None], padding_mask[:, None], bandwidth)
 *     audiocodecs/encodec.py:139-140 self.model.decode(toks[None].movedim(-1, -2), [None])
 *     audiocodecs/encodec.py:116     self.model.encoder(input_values)            (_sig_to_feats)
 *     audiocodecs/encodec.py:125,147 self.model.quantizer.decode(toks)           (_sig_to_qfeats/_toks_to_qfeats)
 *     audiocodecs/encodec.py:74-79   quantizer.layers[k].codebook.embed           (embs)
 * which sit behind Codec.sig_to_toks / toks_to_sig / sig_to_feats / toks_to_qfeats / embs
 * (audiocodecs/codec.py:57-107,182-184).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - plain pointers and sizes only; every *_dev pointer is HIP device memory owned by the caller;
 *  - all work is enqueued on the caller's `stream` (a hipStream_t passed as void*); no entry point
 *    synchronises the device, so the caller's own fences (torch.cuda.synchronize in
 *    downstream/test_sr.py:58,84) time the real work;
 *  - no compute entry point allocates or frees device memory: every byte a call needs beyond the handle's weights
 *    (activations, LSTM state, and the split-operand bookkeeping -- per-clip amax slots and the per-row ring) is carved
 *    from the caller's workspace, whose size ac_encode_workspace_bytes / ac_decode_workspace_bytes /
 *    ac_quantizer_workspace_bytes report for the call's (B, T | N); B may grow or shrink from call to call without any
 *    hipMalloc / hipFree / stream synchronisation inside the library (tests/test_workspace_contract_gpu.py).  The handle
 *    owns only what ac_finalize allocates once: the packed weights, the persistent LSTM's control words, a pinned status
 *    word, and (Mimi) a few KB of pool for ac_embs_projected, the one launching entry point without a workspace argument;
 *  - return 0 on success, a negative AC_E* code on failure; never throws across the ABI;
 *    ac_last_error() returns a human-readable message for the last failure on that handle;
 *  - a handle is not thread-safe; one handle per process/GPU like the reference's one codec/rank;
 *    the handle's device (ac_config.device) must be the current HIP device when its entry points run;
 *  - activations, weights and results are fp32, accumulation is fp32, tokens are int64 like the reference's.  The
 *    large GEMMs, the fused residual blocks and the LSTM products run "split-operand" arithmetic on the fp16 matrix pipe:
 *    every fp32 operand, scaled by a power of two, is written as two fp16 terms and 3 of the 4 exact partial products
 *    are accumulated in fp32 (fp32-grade error: K = 1536 dot products measure 1.9e-7 rms against an fp32 FMA chain's
 *    1.8e-7).  ac_set_precision selects the one alternative: exact fp32 products
 *    (v_mfma_f32_16x16x4_f32) in every kernel.  (The three-bf16-term and rounded-bf16 modes of rounds 1-3 are gone;
 *    their values are rejected.)
 */
#ifndef AUDIOCODECS_AMD_H
#define AUDIOCODECS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AC_OK 0
#define AC_EINVAL (-1)   /* bad argument / shape / unsupported configuration            */
#define AC_ESTATE (-2)   /* call order (weights missing, not finalized, ...)             */
#define AC_ENOMEM (-3)   /* device allocation failed or workspace too small              */
#define AC_EHIP (-4)     /* a HIP runtime call or kernel launch failed                   */
#define AC_ENODEV (-5)   /* no gfx950 device visible                                     */

#define AC_MAX_RATIOS 8

typedef struct ac_handle ac_handle;

/* Mirrors the fields of transformers.EncodecConfig the path depends on.
 * Causal convs, reflect padding, weight-norm (pre-folded), mono, no chunking, normalize=False:
 * the facebook/encodec_24khz variant the reference wrapper loads (encodec.py:49-51). */
typedef struct ac_config {
    int32_t struct_size;               /* = sizeof(ac_config)                                  */
    int32_t sampling_rate;             /* 24000                                                */
    int32_t num_filters;               /* 32                                                   */
    int32_t hidden_size;               /* 128 (latent width == codebook dim)                   */
    int32_t num_ratios;                /* 4                                                    */
    int32_t upsampling_ratios[AC_MAX_RATIOS]; /* 8,5,4,2 (decoder order; encoder uses reverse) */
    int32_t kernel_size;               /* 7                                                    */
    int32_t last_kernel_size;          /* 7                                                    */
    int32_t residual_kernel_size;      /* 3                                                    */
    int32_t compress;                  /* 2                                                    */
    int32_t num_lstm_layers;           /* 2                                                    */
    int32_t codebook_size;             /* 1024                                                 */
    int32_t num_quantizers;            /* 32                                                   */
    int32_t device;                    /* HIP device ordinal                                   */
} ac_config;

/* Mimi: the fields of transformers.MimiConfig the path depends on; defaults = kyutai/mimi,
 * what audiocodecs/mimi.py:45 loads.  Replaces, behind Mimi._sig_to_toks/_toks_to_sig/_sig_to_feats/
 * _toks_to_qfeats/embs (mimi.py:52-156):
 *     mimi.py:105-108  self.model.encode(sig[:, None], padding_mask[:, None], num_quantizers=K)
 *     mimi.py:146-147  self.model.decode(toks.movedim(-1, -2))
 *     mimi.py:115-119  encoder -> encoder_transformer -> downsample
 *     mimi.py:139,153  self.model.quantizer.decode(...)
 * Causal convs with zero ("constant") padding, identity ResBlock shortcuts, no weight-norm, multi-head
 * attention with num_key_value_heads == num_attention_heads, "default" RoPE, exact-erf GELU, LayerScale. */
typedef struct ac_mimi_config {
    int32_t struct_size;               /* = sizeof(ac_mimi_config)                              */
    int32_t sampling_rate;             /* 24000                                                 */
    int32_t num_filters;               /* 64                                                    */
    int32_t hidden_size;               /* 512 (transformer / latent width)                      */
    int32_t num_ratios;                /* 4                                                     */
    int32_t upsampling_ratios[AC_MAX_RATIOS]; /* 8,6,5,4 (decoder order)                        */
    int32_t kernel_size;               /* 7                                                     */
    int32_t last_kernel_size;          /* 3                                                     */
    int32_t residual_kernel_size;      /* 3                                                     */
    int32_t compress;                  /* 2                                                     */
    int32_t codebook_size;             /* 2048                                                  */
    int32_t codebook_dim;              /* 256 (== vector_quantization_hidden_dimension)         */
    int32_t num_quantizers;            /* 32                                                    */
    int32_t num_semantic_quantizers;   /* 1                                                     */
    int32_t num_hidden_layers;         /* 8 (per transformer)                                   */
    int32_t num_attention_heads;       /* 8                                                     */
    int32_t head_dim;                  /* 64                                                    */
    int32_t intermediate_size;         /* 2048                                                  */
    int32_t sliding_window;            /* 250                                                   */
    int32_t resample_stride;           /* 2 (encodec_frame_rate / frame_rate)                   */
    int32_t device;                    /* HIP device ordinal                                    */
    float rope_theta;                  /* 10000                                                 */
    float norm_eps;                    /* 1e-5                                                  */
} ac_mimi_config;

/* DAC (BASELINE.json configs[2]).  The reference wrapper (audiocodecs/dac.py:28-130) calls
 * `dac.DAC` of descript-audio-codec 1.0.0, which is NOT on disk: parity with it is unpinned; this path is
 * pinned to the same-architecture transformers.DacModel (field names below are DacConfig's).  Replaces
 *     dac.py:96-99    self.model.encode(sig[:, None], n_quantizers=K)            -> ac_encode / ac_encode_quantized
 *     dac.py:105-111  self.model.encoder(...), quantizers[0].in_proj(...)          -> ac_encode_feats / ac_encode_feats_latent
 *     dac.py:126-129  self.model.quantizer.from_codes(...), self.model.decode(...) -> ac_decode (ac_dequantize_ws)
 *     dac.py:63-90    codebooks / out_proj(codebooks)                              -> ac_embs / ac_embs_projected */
#define AC_MAX_DILATIONS 4
typedef struct ac_dac_config {
    int32_t struct_size;               /* = sizeof(ac_dac_config)                               */
    int32_t sampling_rate;             /* 44100                                                 */
    int32_t encoder_hidden_size;       /* 64                                                    */
    int32_t decoder_hidden_size;       /* 1536                                                  */
    int32_t num_ratios;                /* 4                                                     */
    int32_t downsampling_ratios[AC_MAX_RATIOS]; /* 2,4,8,8                                      */
    int32_t upsampling_ratios[AC_MAX_RATIOS];   /* 8,8,4,2                                      */
    int32_t n_codebooks;               /* 9                                                     */
    int32_t codebook_size;             /* 1024                                                  */
    int32_t codebook_dim;              /* 8 (the only supported value)                          */
    int32_t num_dilations;             /* 3 residual units per block ...                        */
    int32_t dilations[AC_MAX_DILATIONS]; /* ... with dilations 1,3,9                            */
    int32_t device;
} ac_dac_config;

/* WavTokenizer (BASELINE.json configs[4]).  The reference wrapper (audiocodecs/wavtokenizer.py:31-135)
 * calls the package `wavtokenizer` (lucadellalib/WavTokenizer), which is NOT on disk: PARITY UNPINNED -- this path is
 * pinned to oracle/wavtokenizer_oracle.py, a restatement of the published modules.  Replaces
 *     wavtokenizer.py:94-95    self.model.encode(sig, bandwidth_id=0)                  -> ac_encode (K = 1) / ac_dequantize
 *     wavtokenizer.py:101      self.model.feature_extractor.encodec.encoder(sig[:, None]) -> ac_encode_feats
 *     wavtokenizer.py:115-118  codes_to_features(...) + self.model.decode(feats, bandwidth_id=0) -> ac_decode (ac_dequantize)
 *     wavtokenizer.py:130-133  self.model.decode(feats.movedim(-1,-2), bandwidth_id=0)  -> ac_decode_feats
 *     wavtokenizer.py:87       quantizer.vq.layers[0].codebook                          -> ac_embs
 * Fields: the published YAML configs the wrapper names (:37-40).  Weight names are the checkpoint's own
 * ("feature_extractor.encodec.encoder.model.{i}.conv.conv.{weight_g,weight_v,bias}", "...block.{1,3}...", "...shortcut...",
 * "...model.{i}.lstm.weight_ih_l0", "feature_extractor.encodec.quantizer.vq.layers.0._codebook.embed",
 * "backbone.embed.*", "backbone.pos_net.{0,1,3,4}.{norm1,conv1,norm2,conv2}.*", "backbone.pos_net.2.{norm,q,k,v,proj_out}.*",
 * "backbone.pos_net.5.*", "backbone.norm.{scale,shift}.weight", "backbone.convnext.{l}.{dwconv,norm.scale,norm.shift,
 * pwconv1,pwconv2}.*", "...gamma", "backbone.final_layer_norm.*", "head.out.*", optional "head.istft.window"). */
typedef struct ac_wavtok_config {
    int32_t struct_size;               /* = sizeof(ac_wavtok_config)                                          */
    int32_t sampling_rate;             /* 24000                                                               */
    int32_t num_filters;               /* 32                                                                  */
    int32_t dimension;                 /* 512: encoder output width == codebook dim == backbone input width   */
    int32_t num_ratios;                /* 4                                                                   */
    int32_t ratios[AC_MAX_RATIOS];     /* 6,5,5,4 (`dowmsamples`; the encoder applies them reversed); 75 tok/s: 8,5,4,2 */
    int32_t kernel_size;               /* 7                                                                   */
    int32_t last_kernel_size;          /* 7                                                                   */
    int32_t residual_kernel_size;      /* 3                                                                   */
    int32_t compress;                  /* 2                                                                   */
    int32_t num_lstm_layers;           /* 2                                                                   */
    int32_t codebook_size;             /* 4096                                                                */
    int32_t backbone_dim;              /* 768 (256 also supported)                                            */
    int32_t intermediate_dim;          /* 2304                                                                */
    int32_t num_layers;                /* 12 ConvNeXt blocks                                                  */
    int32_t adanorm_num_embeddings;    /* 4                                                                   */
    int32_t num_groups;                /* 32 (GroupNorm of pos_net)                                           */
    int32_t n_fft;                     /* 2400 (hop 600); 1280 (hop 320); must be a multiple of the hop       */
    int32_t bandwidth_id;              /* 0: the AdaLayerNorm row the wrapper always selects                  */
    int32_t device;
} ac_wavtok_config;

/* Library/ABI version: major*10000 + minor*100 + patch. */
int ac_version(void);

/* Create a handle for `cfg` on device cfg->device.  No device memory is allocated yet. */
int ac_create(const ac_config* cfg, ac_handle** out);

/* Same, for a Mimi handle.  Every other entry point below works on either kind of handle. */
int ac_mimi_create(const ac_mimi_config* cfg, ac_handle** out);

/* Same, for a DAC handle (keys of DacModel.state_dict(): "encoder.conv1.weight", "encoder.block.{i}.
 * res_unit{u}.{snake1.alpha,conv1.weight,...}", "decoder.block.{i}.conv_t1.weight", "quantizer.quantizers.{k}.
 * {in_proj,out_proj}.{weight,bias}", "...codebook.weight"; weights plain, i.e. weight-norm already folded). */
int ac_dac_create(const ac_dac_config* cfg, ac_handle** out);

/* Same, for a WavTokenizer handle. */
int ac_wavtok_create(const ac_wavtok_config* cfg, ac_handle** out);

/* Hand one fp32 tensor to the handle (copied).  `name` uses the HF state-dict keys of
 * EncodecModel with weight-norm either
 *   - already folded:   "<prefix>.weight"  (what the Python host passes; folded with the same torch
 *                        primitive the reference's parametrisation evaluates), or
 *   - unfolded:         "<prefix>.parametrizations.weight.original0" (g) and "...original1" (v);
 *                        folded inside ac_finalize as  w = v * (g / ||v||_2), norm over dims (1,2).
 * plus "<prefix>.bias", "<lstm>.weight_{ih,hh}_l{n}", "<lstm>.bias_{ih,hh}_l{n}",
 * "quantizer.layers.{k}.codebook.embed".  Other keys (embed_avg, cluster_size, inited) are
 * accepted and ignored.  `bytes` must equal 4 * number of elements expected for that key.
 * Mimi handles take the keys of MimiModel.state_dict(): "<conv>.weight"/".bias" (no weight-norm),
 * "{encoder,decoder}_transformer.layers.{l}.{self_attn.{q,k,v,o}_proj,mlp.fc{1,2}}.weight",
 * ".{input,post_attention}_layernorm.{weight,bias}", ".{self_attn,mlp}_layer_scale.scale",
 * "downsample.conv.weight", "upsample.conv.weight",
 * "quantizer.{semantic,acoustic}_residual_vector_quantizer.{input,output}_proj.weight" and
 * "...layers.{q}.codebook.{embed_sum,cluster_usage}" (embed = embed_sum / max(cluster_usage, 1e-5),
 * [HF] mimi :980-983); optionally "encoder_transformer.rotary_emb.inv_freq" (the model's non-persistent
 * buffer; computed as 1/theta^(2i/d) in fp32 when absent). */
int ac_load_weights(ac_handle* h, const char* name, const void* host_ptr, size_t bytes);

/* Arithmetic of the GEMM-shaped kernels; call before ac_finalize (weights are packed for one arithmetic).
 *   AC_PRECISION_FP32        default: fp32 fidelity on the fp16 matrix pipe ("split16", csrc/split16.h): every operand as two
 *                            scaled fp16 planes (x 2^s = hi + lo, both round-to-nearest), 3 partial products, fp32 accumulate;
 *                            power-of-two scales per clip (activations: largest magnitude reported by the producing kernel), per
 *                            row (linear layers over merged token matrices) and per output channel (weights) -- the arithmetic
 *                            every parity claim is made for;
 *   AC_PRECISION_FP32_EXACT  exact fp32 products (v_mfma_f32_16x16x4_f32) everywhere; same as AC_GEMM=fp32;
 * (Rounds 1-3 also carried a three-bf16-plane arithmetic and an opt-in rounded-bf16 side mode; both were removed in round 4: neither
 * had a user, and the side mode kept fp32 activations in HBM -- no parity and no bandwidth saving.  Values 2 and 3 are rejected.)
 * Without this call the environment variable AC_GEMM=fp32 selects the exact-product kernels, default AC_PRECISION_FP32. */
#define AC_PRECISION_FP32 0
#define AC_PRECISION_FP32_EXACT 1
int ac_set_precision(ac_handle* h, int precision);

/* Check that every tensor of the configuration arrived, fold/pack them into the kernels' layouts
 * and upload them (one device allocation owned by the handle). */
int ac_finalize(ac_handle* h);

/* Frames produced for T samples: ceil at every strided conv (T=1..320 -> 1, 321 -> 2, ...). */
int ac_num_frames(const ac_handle* h, int T);
/* Samples ac_decode writes per clip for N frames: N*hop, except DAC (symmetric padding):
 * each transposed conv gives (L-1)*s - 2*ceil(s/2) + 2s.  DAC's ac_num_frames follows the strided convs
 * floor((L + 2*ceil(s/2) - 2s)/s) + 1 and is 0 when the input is too short (upstream's conv1d raises). */
long long ac_num_samples(const ac_handle* h, int N);
/* Hop length (product of ratios, 320; Mimi: x resample_stride = 1920), latent width of feats/qfeats
 * (128; Mimi 512) and codebook vector width (EnCodec: == hidden; Mimi 256). */
int ac_hop_length(const ac_handle* h);
int ac_hidden_size(const ac_handle* h);
int ac_codebook_dim(const ac_handle* h);

/* Scratch the caller must provide (device memory, 256-byte aligned) for one call. */
size_t ac_encode_workspace_bytes(const ac_handle* h, int B, int T);
size_t ac_decode_workspace_bytes(const ac_handle* h, int B, int N);

/* sig_dev [B,T] fp32  ->  toks_dev [B,N,K] int64,  N = ac_num_frames(T).
 * rel_len_dev: NULL, or [B] fp32 relative lengths (SpeechBrain style): sample t of clip b is
 * zeroed before the encoder iff not (float)t < (float)T * rel_len[b]   (encodec.py:84-89,
 * [HF] modeling_encodec.py:589-590).  Tokens are produced for all N frames regardless.
 * K = number of codebooks (quantizer stages), 1 <= K <= num_quantizers.
 * One clip per call is limited to 7 340 031 samples (DAC handles: 3 670 015): per-clip activations are addressed with
 * 32-bit byte offsets; longer clips return AC_EINVAL ("split it") -- the codecs are causal / chunkable on the host. */
int ac_encode(ac_handle* h, const float* sig_dev, const float* rel_len_dev, int B, int T, int K,
              int64_t* toks_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Encoder only: sig_dev [B,T] -> feats_dev [B,N,H] fp32 (channels-last, i.e. the wrapper's
 * `feats.movedim(-1,-2)` layout, encodec.py:116-118).  rel_len_dev as in ac_encode (the reference's
 * _sig_to_feats does not mask for the 24 kHz model: pass NULL to reproduce it). */
int ac_encode_feats(ac_handle* h, const float* sig_dev, const float* rel_len_dev, int B, int T,
                    float* feats_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* toks_dev [B,N,K] int64 (any ids in [0, codebook_size)) -> sig_dev [B, N*hop] fp32.
 * The output is not trimmed to the encoder's input length (encodec.py:139-140). */
int ac_decode(ac_handle* h, const int64_t* toks_dev, int B, int N, int K, float* sig_dev,
              void* workspace_dev, size_t workspace_bytes, void* stream);

/* WavTokenizer only (wavtokenizer.py:128-135 `_feats_to_sig`): feats_dev [B,N,dimension] (channels-last) -> sig_dev [B, N*hop]
 * through backbone + iSTFT head.  Workspace: ac_decode_workspace_bytes(h, B, N). */
int ac_decode_feats(ac_handle* h, const float* feats_dev, int B, int N, float* sig_dev, void* workspace_dev, size_t workspace_bytes,
                    void* stream);

/* DAC only.  ac_encode_quantized: ac_encode that also returns the quantised representation
 * qfeats_dev [B,N,H] `model.encode` yields (dac.py:117-119; not bit-identical to from_codes: the
 * straight-through form rounds).  ac_encode_feats_latent: quantizers[0].in_proj(encoder(sig)) -> [B,N,8]
 * (dac.py:104-108, `latent=True`). */
int ac_encode_quantized(ac_handle* h, const float* sig_dev, int B, int T, int K, int64_t* toks_dev, float* qfeats_dev,
                        void* workspace_dev, size_t workspace_bytes, void* stream);
int ac_encode_feats_latent(ac_handle* h, const float* sig_dev, int B, int T, float* feats_latent_dev,
                           void* workspace_dev, size_t workspace_bytes, void* stream);

/* RVQ only.  ac_quantize: feats_dev [B,N,H] -> toks_dev [B,N,K] ([HF]:424-438).
 * ac_dequantize: toks_dev [B,N,K] -> qfeats_dev [B,N,H] = sum_k E_k[tok]  ([HF]:440-447). */
int ac_quantize(ac_handle* h, const float* feats_dev, int B, int N, int K, int64_t* toks_dev, void* stream);
int ac_dequantize(ac_handle* h, const int64_t* toks_dev, int B, int N, int K, float* qfeats_dev, void* stream);
/* Mimi's split quantiser projects in and out of the codebook space (mimi.py:139,153 ->
 * [HF] mimi :1129-1138), which needs scratch: same calls with a workspace of
 * ac_quantizer_workspace_bytes(h, B, N) bytes (0 for EnCodec handles, which may pass NULL). */
size_t ac_quantizer_workspace_bytes(const ac_handle* h, int B, int N);
int ac_quantize_ws(ac_handle* h, const float* feats_dev, int B, int N, int K, int64_t* toks_dev,
                   void* workspace_dev, size_t workspace_bytes, void* stream);
int ac_dequantize_ws(ac_handle* h, const int64_t* toks_dev, int B, int N, int K, float* qfeats_dev,
                     void* workspace_dev, size_t workspace_bytes, void* stream);

/* Copy the first K codebooks to embs_dev [K, codebook_size, ac_codebook_dim] fp32 (encodec.py:74-79;
 * mimi.py:52-62 `latent=True`). */
int ac_embs(ac_handle* h, int K, float* embs_dev, void* stream);
/* Mimi `latent=False` (mimi.py:63-90): every code vector through its quantiser's output projection
 * -> embs_dev [K, codebook_size, hidden].  AC_EINVAL on EnCodec handles. */
int ac_embs_projected(ac_handle* h, int K, float* embs_dev, void* stream);

/* Sample-rate conversion at the Codec boundary (audiocodecs/codec.py:59-63,95-99 call
 * torchaudio.functional.resample): polyphase windowed-sinc FIR.  kern_dev [n][taps] is the filter
 * bank (n = new_rate/gcd phases, stride o = orig_rate/gcd, `width` zero samples of left padding);
 * y[b][i*n + ph] = sum_k kern[ph][k] * x[b][i*o + k - width], L_out = ceil(n*L/o).  Handle-free. */
int ac_resample(const float* x_dev, int B, int L, const float* kern_dev, int n, int o, int taps, int width,
                float* y_dev, int L_out, void* stream);

/* Optional per-kernel timing with HIP events on the caller's stream (bench.py's roofline leg).
 * ac_profile_begin arms it; every launch made by subsequent calls is bracketed by events.
 * ac_profile_end synchronises those events and writes up to `cap` records; returns the count. */
typedef struct ac_kernel_stat {
    char name[96];       /* kernel (family) name as it appears in rocprofv3 --kernel-trace; with the
                          * environment variable AC_PROF_DETAIL=1 the tap-GEMM records also carry their shape */
    int32_t launches;
    float total_ms;
    double flops;        /* algorithmic flops of those launches (2*M*N*K, ...)             */
    double bytes;        /* algorithmic HBM bytes (inputs read once + outputs written once) */
} ac_kernel_stat;
int ac_profile_begin(ac_handle* h);
int ac_profile_end(ac_handle* h, ac_kernel_stat* out, int cap);

/* Test hook: while armed (buf_dev != NULL), ac_encode/ac_encode_feats/ac_decode append every module
 * output -- in HF module order, standard channels-last [B][L][C] layout -- to buf_dev.
 * ac_debug_captured returns the floats appended so far (may exceed cap_floats: nothing is written
 * past the capacity).  Disarm with ac_debug_capture(h, NULL, 0). */
int ac_debug_capture(ac_handle* h, float* buf_dev, size_t cap_floats);
/* Developer / test switches of a handle: A/B paths whose results are EQUIVALENT (bit-identical or fp32-faithful; named in the
 * parity tests): "tap_epi_staged", "tap_dil", "tap_stagger", "tap_pick", "tap8", "tap8_form", "tap8_spread", "rb_stream",
 * "rb128_stream", "chain_stream", "front_seg", "tail_seg", "front_ldspad", "lstm_fuse_in", "rvq_exact", "prof_detail", "head_seq", "attn_exact",
 * "dac_unit", "mimi_tail".  Their initial values come from the environment variables of the same meaning (AC_TAP_EPI, AC_TAP_DIL,
 * ...), read ONCE, at ac_finalize; no compute entry point reads the environment.
 * "rb6_dbg" (timing modes with WRONG results) and "lstm_dbg" (fault injection, traces) exist in the DEVELOPER library only
 * (libaudiocodecs_amd_dev.so, built beside the product by csrc/build.sh with -DAC_DEVELOPER): the product library returns AC_EINVAL for
 * them, does not read AC_RB6_DBG / AC_LSTM_DBG, and its kernels ignore the words.  Not part of the product interface. */
int ac_debug_set(ac_handle* h, const char* key, int value);
size_t ac_debug_captured(const ac_handle* h);

/* Test hook: the constants of the BOUNDS that stand in for an amax where a tensor exists only inside a fused kernel
 * (csrc/enc_front.h, dec_tail.h, rb_fused6.h): a split16 scale derived from a bound 2^w too large costs w of the 16 bits of
 * range split16 keeps below a tensor's largest element (csrc/split16.h); tests/test_split16_gpu.py measures w on speech-like
 * data.  Writes 11 + 4 * AC_MAX_RATIOS floats (layout at the definition, csrc/ac_api.hip) and returns that count. */
int ac_debug_bounds(const ac_handle* h, float* out, int cap);

/* Diagnostics (SYNCHRONISES): shader clock the tap_gemm6 workgroups ran at since the last call -- every workgroup reads
 * s_memtime (shader clock) and s_memrealtime (100 MHz) at its start and end; *shader_mhz = 100 * sum / sum (0 when nothing ran).
 * enable != 0 arms the sampling for the following calls, 0 disarms it.  The matrix pipe on this chip is power-capped: the
 * clock under a GEMM is the missing half of its roofline. */
int ac_debug_clock(ac_handle* h, int enable, double* shader_mhz);
/* Developer builds only (-DT6_TRACE, tools/experiments/r3o_trace.py): copies the s_memtime stage stamps one tap_gemm6 workgroup
 * left behind the clock words (ac_debug_clock must be enabled); returns the number of 64-bit words written.  SYNCHRONISES. */
int ac_debug_trace(ac_handle* h, unsigned long long* out, int words);

/* Test hook (no GPU): the host-side packer's split of ONE weight row in split16 arithmetic (csrc/split16.h): the row's scale
 * exponent s (|w| 2^s < 2^15, chosen from the row's largest magnitude), and per element the fp16 bit patterns of
 * hi = fp16_rn(w 2^s) and lo = fp16_rn(w 2^s - hi).  Returns s.  The CPU tests compare it with numpy's float16. */
int ac_debug_split_row(const float* w, int n, uint16_t* hi, uint16_t* lo);

/* Which LSTM path the handle uses (SYNCHRONISES the device; tests / diagnostics): 1 = the persistent single-launch kernel
 * (D = 512, 2 layers, 256-CU device; opt out with the environment variable AC_LSTM=step), 0 = one launch per
 * time step, AC_EHIP = a persistent launch failed since the handle was created (a bounded wait expired, or the launch
 * did not get 32 workgroups on every XCD -- e.g. a shared GPU).
 *
 * Failures only the device can see are STICKY and reported by the next entry point called on the handle (no entry point
 * synchronises): the failed call's outputs were set to NaN on the device -- never left unwritten --, the next call
 * returns AC_EHIP once and the handle switches to the per-step LSTM kernels; likewise a token id outside
 * [0, codebook_size) in ac_decode / ac_dequantize sets that frame to NaN and the next call returns AC_EINVAL once
 * (torch.nn.functional.embedding raises).  A clip whose samples contain NaN/Inf does not disturb the other clips of
 * the batch: its LSTM outputs are NaN from that frame on, like the reference's. */
int ac_lstm_status(ac_handle* h);

/* Explicit poll for those sticky device-side failures (round-2 advisor finding: otherwise an unrelated, correct later call is
 * the one that raises).  SYNCHRONISES `stream`, then returns what the next entry point would have returned -- AC_EHIP (a
 * persistent LSTM launch failed; the handle has switched to the per-step kernels), AC_EINVAL (token ids out of range) or AC_OK --
 * and clears the words, so that later calls are not affected.  The Python wrappers call it after each of their own calls when
 * constructed with strict=True. */
int ac_poll_status(ac_handle* h, void* stream);

const char* ac_last_error(const ac_handle* h);
void ac_destroy(ac_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOCODECS_AMD_H */
