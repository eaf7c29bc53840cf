"""CPU oracle for the WavTokenizer path -- TEST INFRASTRUCTURE, not product code.

***PARITY UNPINNED.***  The reference wrapper (/root/reference/audiocodecs/wavtokenizer.py:31-135) calls the
third-party package `wavtokenizer` (`pip install git+https://github.com/lucadellalib/WavTokenizer.git@main`,
requirements.txt:21, unpinned), which is NOT installed and NOT under /root/reference: `import wavtokenizer` fails, so
the reference cannot run here and no golden vector of it exists.  This file restates the PUBLISHED algorithm of that
package (a fork of jishengpeng/WavTokenizer: decoder/pretrained.py, decoder/feature_extractors.py, decoder/models.py,
decoder/modules.py, decoder/heads.py, decoder/spectral_ops.py, encoder/modules/{seanet,conv,lstm}.py,
encoder/quantization/core_vq.py) as recalled, anchored on the wrapper's own call sites:

    wavtokenizer.py:94-95    self.model.encode(sig, bandwidth_id=0) -> (features, codes[K=1,B,N])     sig_to_toks / sig_to_qfeats
    wavtokenizer.py:101      self.model.feature_extractor.encodec.encoder(sig[:, None])                 sig_to_feats
    wavtokenizer.py:115-118  codes_to_features(toks.movedim(-1,0)); decode(feats, bandwidth_id=tensor(0)) toks_to_sig
    wavtokenizer.py:87       feature_extractor.encodec.quantizer.vq.layers[0].codebook                   embs  [1,4096,512]
    wavtokenizer.py:130      self.model.decode(feats.movedim(-1,-2), bandwidth_id=0)                     feats_to_sig

What the restated modules do (eval mode):
  * SEANetEncoder(causal=False, pad_mode="reflect", norm="weight_norm", lstm=2, n_residual_layers=1, true_skip=False,
    ratios=dowmsamples (applied reversed), dimension=512): SConv1d pads  padding_total = (k-1)*dil+1 - stride  as
    right = total//2 (+ extra padding that completes the last frame), left = total - right, reflect, with the
    small-input rule (zero-extend to max_pad+1, reflect, drop);
  * ResidualVectorQuantizer with ONE EuclideanCodebook: dist = -(|x|^2 - 2 x.E^T + |E|^2), argmax; quantized = E[idx];
  * VocosBackbone: Conv1d(512,768,k7,pad 3) -> pos_net [ResnetBlock x2, AttnBlock, ResnetBlock x2, GroupNorm(32, eps 1e-6)]
    -> AdaLayerNorm(cond 0, eps 1e-6) -> 12 ConvNeXt blocks (dwconv k7, AdaLayerNorm, Linear 768->2304, GELU, Linear
    2304->768, gamma, residual) -> LayerNorm(eps 1e-6);
    ResnetBlock: x + conv2(swish(GN(conv1(swish(GN(x))))))  (k3, zero pad 1; dropout is identity in eval);
    AttnBlock: x + proj_out(softmax(q^T k / sqrt(C)) applied to v), single head over all frames;
  * ISTFTHead: Linear(768, n_fft+2) -> (mag, phase) halves; mag = min(exp(mag), 100); S = mag*(cos p + i sin p);
    ISTFT(padding="same"): irfft(n_fft) * hann window, overlap-add (fold), trim (win-hop)/2 both sides, divide by the
    overlap-added squared window.
"""

from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from oracle.encodec_oracle import lstm_skip


def cast_weights(sd: Dict[str, torch.Tensor], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Checkpoint -> plain tensors: folds old-style weight-norm (weight_g, weight_v) of the encoder convs with the
    primitive torch.nn.utils.weight_norm evaluates."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            p = k[: -len(".weight_g")]
            out[p + ".weight"] = torch._weight_norm(sd[p + ".weight_v"].to(dtype), v.to(dtype), 0)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v.to(dtype) if v.is_floating_point() else v
    return out


# --------------------------------------------------------------------------------------------- encoder
def pad1d(x, left: int, right: int):
    """encoder/modules/conv.py pad1d(mode="reflect") incl. the small-input rule."""
    length = x.shape[-1]
    max_pad = max(left, right)
    extra = 0
    if length <= max_pad:
        extra = max_pad - length + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (left, right), mode="reflect")
    return y[..., : y.shape[-1] - extra]


def sconv1d(x, w, b, stride: int = 1):
    """SConv1d.forward, causal=False (dilation 1 everywhere in this model)."""
    k = w.shape[-1]
    total = k - stride
    length = x.shape[-1]
    n_frames = (length - k + total) / stride + 1
    ideal = (math.ceil(n_frames) - 1) * stride + (k - total)
    extra = ideal - length
    right = total // 2
    left = total - right
    return F.conv1d(pad1d(x, left, right + extra), w, b, stride=stride)


def resblock(x, W, p: str):
    h = sconv1d(F.elu(x), W[p + ".block.1.conv.conv.weight"], W[p + ".block.1.conv.conv.bias"])
    h = sconv1d(F.elu(h), W[p + ".block.3.conv.conv.weight"], W[p + ".block.3.conv.conv.bias"])
    return sconv1d(x, W[p + ".shortcut.conv.conv.weight"], W[p + ".shortcut.conv.conv.bias"]) + h


def encoder(cfg, W, x, taps: Optional[dict] = None):
    """SEANetEncoder: x [B,1,T] -> [B,dimension,N]."""
    p = "feature_extractor.encodec.encoder.model."
    h = sconv1d(x, W[p + "0.conv.conv.weight"], W[p + "0.conv.conv.bias"])
    if taps is not None:
        taps["enc0"] = h
    i = 1
    for r in reversed(cfg.ratios):
        h = resblock(h, W, f"{p}{i}")
        if taps is not None:
            taps[f"enc{i}"] = h
        h = sconv1d(F.elu(h), W[f"{p}{i + 2}.conv.conv.weight"], W[f"{p}{i + 2}.conv.conv.bias"], stride=r)
        if taps is not None:
            taps[f"enc{i + 2}"] = h
        i += 3
    h = lstm_skip(h, W, f"{p}{i}.lstm", cfg.num_lstm_layers)
    if taps is not None:
        taps[f"enc{i}"] = h
    h = sconv1d(F.elu(h), W[f"{p}{i + 2}.conv.conv.weight"], W[f"{p}{i + 2}.conv.conv.bias"])
    return h


# --------------------------------------------------------------------------------------------- quantiser
def codebook(W):
    return W["feature_extractor.encodec.quantizer.vq.layers.0._codebook.embed"]


def vq_encode(E, z, return_margin: bool = False):
    """EuclideanCodebook.quantize: z [B,D,N] -> idx [B,N] (first maximum on ties, torch.max)."""
    B, D, N = z.shape
    x = z.permute(0, 2, 1).reshape(-1, D)
    et = E.t()
    dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ et + et.pow(2).sum(0, keepdim=True))
    idx = dist.max(dim=-1).indices.view(B, N)
    if return_margin:
        top2 = dist.topk(2, dim=-1).values
        return idx, ((top2[:, 0] - top2[:, 1]) / top2[:, 0].abs().clamp_min(1e-30)).view(B, N)
    return idx


# --------------------------------------------------------------------------------------------- backbone
def _swish(x):
    return x * torch.sigmoid(x)


def _gn(x, W, p: str, groups: int):
    return F.group_norm(x, groups, W[p + ".weight"], W[p + ".bias"], eps=1e-6)


def resnet_block(x, W, p: str, groups: int):
    h = F.conv1d(_swish(_gn(x, W, p + ".norm1", groups)), W[p + ".conv1.weight"], W[p + ".conv1.bias"], padding=1)
    h = F.conv1d(_swish(_gn(h, W, p + ".norm2", groups)), W[p + ".conv2.weight"], W[p + ".conv2.bias"], padding=1)
    return x + h


def attn_block(x, W, p: str, groups: int):
    h = _gn(x, W, p + ".norm", groups)
    q = F.conv1d(h, W[p + ".q.weight"], W[p + ".q.bias"])
    k = F.conv1d(h, W[p + ".k.weight"], W[p + ".k.bias"])
    v = F.conv1d(h, W[p + ".v.weight"], W[p + ".v.bias"])
    c = q.shape[1]
    w_ = torch.bmm(q.permute(0, 2, 1), k) * (int(c) ** (-0.5))     # [B, Nq, Nk]
    w_ = F.softmax(w_, dim=2)
    h = torch.bmm(v, w_.permute(0, 2, 1))                           # [B, C, Nq]
    return x + F.conv1d(h, W[p + ".proj_out.weight"], W[p + ".proj_out.bias"])


def _adanorm(x, W, p: str, cond: int):
    """AdaLayerNorm over the last dim: layer_norm (no affine, eps 1e-6) * scale[cond] + shift[cond]."""
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6) * W[p + ".scale.weight"][cond] + W[p + ".shift.weight"][cond]


def convnext(x, W, p: str, cond: int):
    h = F.conv1d(x, W[p + ".dwconv.weight"], W[p + ".dwconv.bias"], padding=3, groups=x.shape[1]).transpose(1, 2)
    h = _adanorm(h, W, p + ".norm", cond)
    h = F.linear(h, W[p + ".pwconv1.weight"], W[p + ".pwconv1.bias"])
    h = F.gelu(h)
    h = F.linear(h, W[p + ".pwconv2.weight"], W[p + ".pwconv2.bias"])
    h = W[p + ".gamma"] * h
    return x + h.transpose(1, 2)


def backbone(cfg, W, feats, taps: Optional[dict] = None):
    """VocosBackbone.forward: feats [B,dimension,N] -> [B,N,backbone_dim]."""
    g, cond = cfg.num_groups, cfg.bandwidth_id
    x = F.conv1d(feats, W["backbone.embed.weight"], W["backbone.embed.bias"], padding=3)
    if taps is not None:
        taps["embed"] = x
    for i in (0, 1):
        x = resnet_block(x, W, f"backbone.pos_net.{i}", g)
        if taps is not None:
            taps[f"pos{i}"] = x
    x = attn_block(x, W, "backbone.pos_net.2", g)
    if taps is not None:
        taps["pos2"] = x
    for i in (3, 4):
        x = resnet_block(x, W, f"backbone.pos_net.{i}", g)
        if taps is not None:
            taps[f"pos{i}"] = x
    x = _gn(x, W, "backbone.pos_net.5", g)
    if taps is not None:
        taps["pos5"] = x
    x = _adanorm(x.transpose(1, 2), W, "backbone.norm", cond).transpose(1, 2)
    if taps is not None:
        taps["norm"] = x
    for l in range(cfg.num_layers):
        x = convnext(x, W, f"backbone.convnext.{l}", cond)
        if taps is not None:
            taps[f"cnx{l}"] = x
    x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), W["backbone.final_layer_norm.weight"], W["backbone.final_layer_norm.bias"], eps=1e-6)
    if taps is not None:
        taps["final"] = x.transpose(1, 2)
    return x


def istft_same(spec, n_fft: int, hop: int, window):
    """ISTFT.forward with padding="same": spec [B, n_fft/2+1, N] complex -> [B, N*hop]."""
    pad = (n_fft - hop) // 2
    B, _, N = spec.shape
    ifft = torch.fft.irfft(spec, n_fft, dim=1, norm="backward") * window[None, :, None]
    out_size = (N - 1) * hop + n_fft
    y = F.fold(ifft, output_size=(1, out_size), kernel_size=(1, n_fft), stride=(1, hop))[:, 0, 0, pad:-pad]
    wsq = window.square().expand(1, N, -1).transpose(1, 2)
    env = F.fold(wsq, output_size=(1, out_size), kernel_size=(1, n_fft), stride=(1, hop)).squeeze()[pad:-pad]
    assert bool((env > 1e-11).all())
    return y / env


def head(cfg, W, x, taps: Optional[dict] = None):
    """ISTFTHead.forward: x [B,N,backbone_dim] -> [B, N*hop]."""
    y = F.linear(x, W["head.out.weight"], W["head.out.bias"]).transpose(1, 2)
    mag, p = y.chunk(2, dim=1)
    mag = torch.clip(torch.exp(mag), max=1e2)
    spec = mag * (torch.cos(p) + 1j * torch.sin(p))
    if taps is not None:
        taps["spec_re"], taps["spec_im"] = spec.real, spec.imag
    window = W.get("head.istft.window")
    if window is None:
        window = torch.hann_window(cfg.n_fft, dtype=x.dtype)
    return istft_same(spec, cfg.n_fft, cfg.hop_length, window.to(x.dtype))


# --------------------------------------------------------------------------------------------- wrapper-level entry points
def sig_to_feats(cfg, W, sig, taps=None):
    """wavtokenizer.py:99-103 -> [B,N,dimension]."""
    return encoder(cfg, W, sig[:, None].to(codebook(W).dtype), taps).movedim(-1, -2)


def sig_to_toks(cfg, W, sig, return_margin: bool = False):
    """wavtokenizer.py:92-96: model.encode(sig, bandwidth_id=0)[1].movedim(0,-1) -> [B,N,1] int64."""
    z = encoder(cfg, W, sig[:, None].to(codebook(W).dtype))
    r = vq_encode(codebook(W), z, return_margin)
    if return_margin:
        return r[0][..., None], r[1][..., None]
    return r[..., None]


def toks_to_qfeats(cfg, W, toks):
    """wavtokenizer.py:121-126: codes_to_features (an embedding lookup; offsets are 0 for the single codebook) -> [B,N,dimension]."""
    return F.embedding(toks[..., 0], codebook(W))


def sig_to_qfeats(cfg, W, sig):
    """wavtokenizer.py:106-110: the quantised features `model.encode` returns = codebook[idx] in eval mode."""
    return toks_to_qfeats(cfg, W, sig_to_toks(cfg, W, sig))


def feats_to_sig(cfg, W, feats, taps=None):
    """wavtokenizer.py:128-135: model.decode(feats.movedim(-1,-2), bandwidth_id=0) -> [B, N*hop]."""
    return head(cfg, W, backbone(cfg, W, feats.movedim(-1, -2), taps), taps)


def toks_to_sig(cfg, W, toks, taps=None):
    """wavtokenizer.py:112-119."""
    return feats_to_sig(cfg, W, toks_to_qfeats(cfg, W, toks), taps)


def embs(W):
    """wavtokenizer.py:84-89: [1, codebook_size, dimension]."""
    return codebook(W)[None]
