"""ORACLE -- test infrastructure only; never imported by the product path.

CPU restatement (torch functional ops, dtype-generic: float32 or float64) of the Mimi encode/decode
path the reference wrapper runs:

    audiocodecs.Codec.sig_to_toks -> Mimi._sig_to_toks    /root/reference/audiocodecs/mimi.py:93-109
    audiocodecs.Codec.toks_to_sig -> Mimi._toks_to_sig    mimi.py:143-148
    Mimi._sig_to_feats / _sig_to_qfeats / _toks_to_qfeats / embs   mimi.py:112-141,151-156,52-90

The arithmetic lives in a third-party dependency that is NOT under /root/reference: `transformers`
(pinned 4.46.3 in downstream/environment.yml:253; 5.15.0 in this image), file
``models/mimi/modeling_mimi.py`` -- cited below as [HF]:line (5.15.0 line numbers).

PARITY PIN: checked against the reference wrapper itself (``audiocodecs.mimi.Mimi`` imported from
/root/reference in the build container, seeded synthetic weights -- tools/make_golden_mimi.py)
through tests/golden/mimi_golden.npz; see tests/test_mimi_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


def _cfg_get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


def cast_weights(sd: Dict[str, torch.Tensor], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """HF-format state dict -> working dtype.  Codebooks are materialised the way
    MimiEuclideanCodebook.embed does ([HF]:980-983): embed_sum / clamp(cluster_usage, 1e-5), in fp32
    (the model's own dtype) and only then cast."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".embed_sum"):
            p = k[: -len(".embed_sum")]
            e = v.float() / sd[p + ".cluster_usage"].float().clamp(min=1e-5)[:, None]
            out[p + ".embed"] = e.to(dtype)
        elif k.endswith(".cluster_usage") or k.endswith(".initialized"):
            continue
        else:
            out[k] = v.to(dtype) if v.is_floating_point() else v
    return out


# --------------------------------------------------------------------------------------------
# SEANet layers
# --------------------------------------------------------------------------------------------


def conv1d_causal(x, w, b, stride: int = 1, pad_mode: str = "constant"):
    """[HF]:318-344 MimiConv1d.forward, causal branch: left pad = kernel - stride, right pad = the
    extra padding that makes the last window full (:266-277); `pad_mode` "constant" (zeros) for every
    SEANet conv, "replicate" for the down-sampler (:1199-1208).  Dilation is 1 everywhere in the
    configured model (num_residual_layers = 1 -> dilation_growth_rate**0)."""
    k = w.shape[-1]
    length = x.shape[-1]
    padding_total = k - stride
    n_frames = math.ceil((length - k + padding_total) / stride + 1) - 1
    extra = n_frames * stride + k - padding_total - length
    x = F.pad(x, (padding_total, extra), mode=pad_mode)
    return F.conv1d(x, w, b, stride=stride)


def convtr1d_causal(x, w, b, stride: int, groups: int = 1):
    """[HF]:347-400 MimiConvTranspose1d: conv_transpose1d then trim kernel - stride samples on the
    right (causal, trim_right_ratio = 1.0)."""
    k = w.shape[-1]
    y = F.conv_transpose1d(x, w, b, stride=stride, groups=groups)
    return y[..., : y.shape[-1] - (k - stride)]


def resblock(x, W, p: str):
    """[HF]:403-441 MimiResnetBlock: x + conv_k1(ELU(conv_k3(ELU(x)))), identity shortcut."""
    h = conv1d_causal(F.elu(x), W[p + ".block.1.conv.weight"], W[p + ".block.1.conv.bias"])
    h = conv1d_causal(F.elu(h), W[p + ".block.3.conv.weight"], W[p + ".block.3.conv.bias"])
    return x + h


def encoder(cfg, W, x, taps: Optional[dict] = None):
    """[HF]:444-492 MimiEncoder: x [B,1,T] -> [B,hidden,T/prod(ratios)]."""
    ratios = list(_cfg_get(cfg, "upsampling_ratios"))

    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    x = tap("encoder.layers.0", conv1d_causal(x, W["encoder.layers.0.conv.weight"], W["encoder.layers.0.conv.bias"]))
    i = 1
    for r in reversed(ratios):
        x = tap(f"encoder.layers.{i}", resblock(x, W, f"encoder.layers.{i}"))
        p = f"encoder.layers.{i + 2}.conv"
        x = tap(f"encoder.layers.{i + 2}", conv1d_causal(F.elu(x), W[p + ".weight"], W[p + ".bias"], stride=r))
        i += 3
    p = f"encoder.layers.{i + 1}.conv"
    return tap(f"encoder.layers.{i + 1}", conv1d_causal(F.elu(x), W[p + ".weight"], W[p + ".bias"]))


def decoder(cfg, W, z, taps: Optional[dict] = None):
    """[HF]:931-961 MimiDecoder: z [B,hidden,N'] -> [B,1,N'*prod(ratios)]."""
    ratios = list(_cfg_get(cfg, "upsampling_ratios"))

    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    x = tap("decoder.layers.0", conv1d_causal(z, W["decoder.layers.0.conv.weight"], W["decoder.layers.0.conv.bias"]))
    i = 1
    for r in ratios:
        p = f"decoder.layers.{i + 1}.conv"
        x = tap(f"decoder.layers.{i + 1}", convtr1d_causal(F.elu(x), W[p + ".weight"], W[p + ".bias"], r))
        x = tap(f"decoder.layers.{i + 2}", resblock(x, W, f"decoder.layers.{i + 2}"))
        i += 3
    p = f"decoder.layers.{i + 1}.conv"
    return tap(f"decoder.layers.{i + 1}", conv1d_causal(F.elu(x), W[p + ".weight"], W[p + ".bias"]))


# --------------------------------------------------------------------------------------------
# transformer
# --------------------------------------------------------------------------------------------


def rope_tables(cfg, T: int, dtype):
    """[HF]:528-567 MimiRotaryEmbedding ("default" rope): inv_freq = 1 / theta^(2i/d) and the
    position products are evaluated in fp32 whatever the model dtype, then cast."""
    d = _cfg_get(cfg, "head_dim")
    inv_freq = 1.0 / (_cfg_get(cfg, "rope_theta") ** (torch.arange(0, d, 2, dtype=torch.float) / d))
    freqs = (inv_freq[:, None] @ torch.arange(T, dtype=torch.float)[None, :]).transpose(0, 1)  # [T, d/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def transformer(cfg, W, x, part: str, taps: Optional[dict] = None):
    """[HF]:729-928 MimiTransformerModel on x [B,T,hidden]: per layer
        x = x + scale_a * o_proj(attn(LN(x)));   x = x + scale_m * fc2(gelu(fc1(LN(x))))
    attention ([HF]:657-727): RoPE on q,k, causal sliding-window mask (key j visible to query i iff
    j <= i and i - j < sliding_window), softmax in fp32, scaling 1/sqrt(head_dim)."""
    B, T, H = x.shape
    nh, hd = _cfg_get(cfg, "num_attention_heads"), _cfg_get(cfg, "head_dim")
    eps, win = _cfg_get(cfg, "norm_eps"), _cfg_get(cfg, "sliding_window")
    cos, sin = rope_tables(cfg, T, x.dtype)
    i = torch.arange(T)
    visible = (i[None, :] <= i[:, None]) & (i[:, None] - i[None, :] < win)
    bias = torch.zeros(T, T, dtype=x.dtype).masked_fill(~visible, float("-inf"))
    for l in range(_cfg_get(cfg, "num_hidden_layers")):
        p = f"{part}.layers.{l}"
        h = F.layer_norm(x, (H,), W[p + ".input_layernorm.weight"], W[p + ".input_layernorm.bias"], eps)
        q = F.linear(h, W[p + ".self_attn.q_proj.weight"]).view(B, T, nh, hd).transpose(1, 2)
        k = F.linear(h, W[p + ".self_attn.k_proj.weight"]).view(B, T, nh, hd).transpose(1, 2)
        v = F.linear(h, W[p + ".self_attn.v_proj.weight"]).view(B, T, nh, hd).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        a = torch.matmul(q, k.transpose(2, 3)) * (1.0 / math.sqrt(hd)) + bias
        a = F.softmax(a, dim=-1, dtype=torch.float32 if x.dtype == torch.float32 else x.dtype).to(x.dtype)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, nh * hd)
        o = F.linear(o, W[p + ".self_attn.o_proj.weight"])
        x = x + W[p + ".self_attn_layer_scale.scale"] * o
        h = F.layer_norm(x, (H,), W[p + ".post_attention_layernorm.weight"], W[p + ".post_attention_layernorm.bias"], eps)
        h = F.linear(F.gelu(F.linear(h, W[p + ".mlp.fc1.weight"])), W[p + ".mlp.fc2.weight"])
        x = x + W[p + ".mlp_layer_scale.scale"] * h
        if taps is not None:
            taps[p] = x
    return x


# --------------------------------------------------------------------------------------------
# split residual vector quantiser
# --------------------------------------------------------------------------------------------

_PARTS = ("semantic", "acoustic")


def _rvq(part: str) -> str:
    return f"quantizer.{part}_residual_vector_quantizer"


def codebooks(cfg, W, K: int) -> List[torch.Tensor]:
    """(semantic_layers + acoustic_layers)[:K] (mimi.py:54-61)."""
    nsem = _cfg_get(cfg, "num_semantic_quantizers")
    out = []
    for q in range(K):
        part, idx = ("semantic", q) if q < nsem else ("acoustic", q - nsem)
        out.append(W[f"{_rvq(part)}.layers.{idx}.codebook.embed"])
    return out


def nearest_code(e, r, return_margin: bool = False):
    """[HF]:985-990 MimiEuclideanCodebook.quantize: argmin over torch.cdist(x, embed, p=2) -- the
    EUCLIDEAN distance (a square root of the clamped squared distance), first index on ties.
    r [F,D], e [C,D].  margin = (d2 - d1) / d2 of the two smallest distances."""
    dist = torch.cdist(r[None], e[None], p=2)[0]
    idx = dist.argmin(dim=-1)
    if not return_margin:
        return idx
    two = torch.topk(dist, 2, dim=-1, largest=False).values
    return idx, (two[:, 1] - two[:, 0]) / two[:, 1].clamp_min(1e-30)


def rvq_encode(cfg, W, z, K: int, return_margin: bool = False):
    """[HF]:1084-1127 MimiSplitResidualVectorQuantizer.encode on z [B,hidden,N] -> codes [K,B,N].
    The semantic RVQ (1 stage) and the acoustic RVQ (K-1 stages) BOTH start from the embedding, each
    through its own 1x1 input projection (:1052-1053); residual -= embed[idx] per stage (:1057-1062)."""
    nsem = _cfg_get(cfg, "num_semantic_quantizers")
    nq = _cfg_get(cfg, "num_quantizers")
    if K > nq:
        raise ValueError(
            f"The number of quantizers (i.e codebooks) asked should be lower than the total number of quantizers {nq}, but is currently {K}."
        )
    if K < nsem:
        raise ValueError(
            f"The number of quantizers (i.e codebooks) asked should be higher than the number of semantic quantizers {nsem}, but is currently {K}."
        )
    B, _, N = z.shape
    codes, margins = [], []
    for part, n in (("semantic", nsem), ("acoustic", K - nsem)):
        if n == 0:
            continue
        r = F.conv1d(z, W[_rvq(part) + ".input_proj.weight"]).permute(0, 2, 1).reshape(B * N, -1)
        for q in range(n):
            e = W[f"{_rvq(part)}.layers.{q}.codebook.embed"]
            res = nearest_code(e, r, return_margin)
            idx = res[0] if return_margin else res
            if return_margin:
                margins.append(res[1].view(B, N))
            r = r - e[idx]
            codes.append(idx.view(B, N))
    codes = torch.stack(codes)
    return (codes, torch.stack(margins)) if return_margin else codes


def rvq_decode(cfg, W, codes):
    """[HF]:1129-1138 split decode of codes [B,K,N]: output_proj_s(sum of semantic code vectors) +
    output_proj_a(sum of acoustic code vectors); each sum starts from 0.0 in stage order (:1068-1081)."""
    nsem = _cfg_get(cfg, "num_semantic_quantizers")
    K = codes.shape[1]
    out = None
    for part, lo, hi in (("semantic", 0, nsem), ("acoustic", nsem, K)):
        if hi <= lo:
            continue
        q = None
        for j in range(lo, hi):
            e = W[f"{_rvq(part)}.layers.{j - lo}.codebook.embed"]
            v = F.embedding(codes[:, j], e).permute(0, 2, 1)  # [B,D,N]
            q = v if q is None else q + v
        q = F.conv1d(q, W[_rvq(part) + ".output_proj.weight"])
        out = q if out is None else out + q
    return out


# --------------------------------------------------------------------------------------------
# the wrapper entry points
# --------------------------------------------------------------------------------------------


def _dtype(W):
    return W["encoder.layers.0.conv.bias"].dtype


def embeddings(cfg, W, sig, taps: Optional[dict] = None):
    """[HF]:1237-1259 `_encode_frame` up to the quantiser: encoder -> encoder_transformer ->
    downsample (replicate-padded stride-2 conv, no bias).  The padding mask the wrapper builds
    (mimi.py:95-104) is NOT applied to the samples ([HF]:1245-1247 leave it unused): `length` has
    no effect on Mimi's outputs."""
    x = encoder(cfg, W, sig[:, None].to(_dtype(W)), taps)
    x = transformer(cfg, W, x.transpose(1, 2), "encoder_transformer", taps).transpose(1, 2)
    z = conv1d_causal(x, W["downsample.conv.weight"], None, stride=_cfg_get(cfg, "resample_stride"), pad_mode="replicate")
    if taps is not None:
        taps["downsample"] = z
    return z


def sig_to_feats(cfg, W, sig, length=None, taps=None):
    """mimi.py:112-121 -> [B,N,hidden]."""
    return embeddings(cfg, W, sig, taps).movedim(-1, -2)


def sig_to_toks(cfg, W, sig, length=None, num_codebooks: int = 8, return_margin: bool = False):
    """mimi.py:93-109: model.encode(sig[:,None], mask, num_quantizers=K).audio_codes.movedim(-1,-2)
    -> [B,N,K] int64."""
    z = embeddings(cfg, W, sig)
    res = rvq_encode(cfg, W, z, num_codebooks, return_margin)
    if return_margin:
        return res[0].permute(1, 2, 0).contiguous(), res[1].permute(1, 2, 0).contiguous()
    return res.permute(1, 2, 0).contiguous()


def toks_to_qfeats(cfg, W, toks):
    """mimi.py:151-156: quantizer.decode(toks.movedim(-1,-2)).movedim(-1,-2) -> [B,N,hidden]."""
    return rvq_decode(cfg, W, toks.movedim(-1, -2)).movedim(-1, -2)


def sig_to_qfeats(cfg, W, sig, length=None, num_codebooks: int = 8):
    """mimi.py:124-141."""
    return toks_to_qfeats(cfg, W, sig_to_toks(cfg, W, sig, length, num_codebooks))


def toks_to_sig(cfg, W, toks, taps=None):
    """mimi.py:143-148 -> [HF]:1399-1414 `_decode_frame`: quantizer.decode -> upsample (depthwise
    transposed conv, stride 2, no bias) -> decoder_transformer -> decoder.  Output [B, N*hop], not
    trimmed (no padding_mask is passed)."""
    z = rvq_decode(cfg, W, toks.movedim(-1, -2))
    if taps is not None:
        taps["quantizer.decode"] = z
    w = W["upsample.conv.weight"]
    x = convtr1d_causal(z, w, None, _cfg_get(cfg, "resample_stride"), groups=w.shape[0])
    if taps is not None:
        taps["upsample"] = x
    x = transformer(cfg, W, x.transpose(1, 2), "decoder_transformer", taps).transpose(1, 2)
    return decoder(cfg, W, x, taps)[:, 0]


def embs(cfg, W, num_codebooks: int, latent: bool = True):
    """mimi.py:52-90: stacked codebooks [K,C,D]; with latent=False each codebook is pushed through
    its quantiser's 1x1 output projection -> [K,C,hidden]."""
    cbs = torch.stack(codebooks(cfg, W, num_codebooks))
    if latent:
        return cbs
    nsem = _cfg_get(cfg, "num_semantic_quantizers")
    out = []
    for q in range(num_codebooks):
        part = "semantic" if q < nsem else "acoustic"
        out.append(F.linear(cbs[q], W[_rvq(part) + ".output_proj.weight"][..., 0]))
    return torch.stack(out)
