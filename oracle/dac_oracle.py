"""ORACLE -- test infrastructure only; never imported by the product path.

CPU restatement (torch functional ops, float32 or float64) of the DAC encode/decode path the reference
wrapper runs:

    audiocodecs.Codec.sig_to_toks -> DAC._sig_to_toks   /root/reference/audiocodecs/dac.py:93-100
    audiocodecs.Codec.toks_to_sig -> DAC._toks_to_sig   dac.py:123-130
    DAC._sig_to_feats / _sig_to_qfeats / embs           dac.py:103-120, 63-90

PARITY UNPINNED with respect to the reference's own backend: the arithmetic lives in the third-party
package `descript-audio-codec==1.0.0` (downstream/environment.yml:71), which is NOT installed here and not
under /root/reference (`import dac` fails, dac.py:44-48), so the reference wrapper cannot be run in this
container.  What is restated below is that package's published algorithm (dac/model/dac.py,
dac/nn/layers.py, dac/nn/quantize.py) as the wrapper calls it; it is PINNED ONLY TO A STAND-IN: the
same-architecture third-party `transformers.DacModel` ([HF] = models/dac/modeling_dac.py, 5.15.0 line
numbers), called the way the wrapper calls `dac.DAC` (tools/make_golden_dac.py ->
tests/golden/dac_golden.npz, tests/test_dac_oracle_golden.py).  One known deviation of the stand-in is
kept switchable (`variant`): descript evaluates the code search as  (-dist).max  with
dist = |e|^2 - 2 e.c + |c|^2, [HF]:167-168 as  max(-(|e|^2 - 2 e.c) + |c|^2)  -- the sign of the (unit) codebook
norm differs, which can only move exact near-ties.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


def _cfg_get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


def cast_weights(sd: Dict[str, torch.Tensor], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


# --------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------


def snake(x, alpha):
    """[HF]:95-100 (dac/nn/layers.py `snake`): x + (alpha + 1e-9)^-1 * sin(alpha*x)^2, alpha [1,C,1]."""
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def res_unit(x, W, p: str, dilation: int):
    """[HF]:175-209 DacResidualUnit: x + conv_k1(snake(conv_k7,dilated(snake(x)))), symmetric zero padding
    3*dilation so the length is kept (the crop at :204-206 is a no-op then)."""
    y = F.conv1d(snake(x, W[p + ".snake1.alpha"]), W[p + ".conv1.weight"], W[p + ".conv1.bias"], dilation=dilation, padding=3 * dilation)
    y = F.conv1d(snake(y, W[p + ".snake2.alpha"]), W[p + ".conv2.weight"], W[p + ".conv2.bias"])
    return x + y


def encoder(cfg, W, x, taps: Optional[dict] = None):
    """[HF]:444-474 DacEncoder: conv k7 pad 3; per stride s: 3 residual units (dilations 1,3,9), snake,
    conv k=2s stride s pad ceil(s/2) ([HF]:212-233); snake; conv k3 pad 1.  x [B,1,T] -> [B,hidden,N]."""
    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    x = tap("encoder.conv1", F.conv1d(x, W["encoder.conv1.weight"], W["encoder.conv1.bias"], padding=3))
    for i, s in enumerate(_cfg_get(cfg, "downsampling_ratios")):
        p = f"encoder.block.{i}"
        for u, d in enumerate(_cfg_get(cfg, "dilations"), start=1):
            x = tap(f"{p}.res_unit{u}", res_unit(x, W, f"{p}.res_unit{u}", d))
        x = tap(f"{p}.conv1", F.conv1d(snake(x, W[p + ".snake1.alpha"]), W[p + ".conv1.weight"], W[p + ".conv1.bias"],
                                       stride=s, padding=math.ceil(s / 2)))
    return tap("encoder.conv2", F.conv1d(snake(x, W["encoder.snake1.alpha"]), W["encoder.conv2.weight"], W["encoder.conv2.bias"], padding=1))


def decoder(cfg, W, z, taps: Optional[dict] = None):
    """[HF]:407-441 DacDecoder: conv k7 pad 3; per stride s: snake, transposed conv k=2s stride s pad
    ceil(s/2), 3 residual units ([HF]:236-264); snake; conv k7 pad 3; tanh.  z [B,hidden,N] -> [B,1,T']."""
    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    x = tap("decoder.conv1", F.conv1d(z, W["decoder.conv1.weight"], W["decoder.conv1.bias"], padding=3))
    for i, s in enumerate(_cfg_get(cfg, "upsampling_ratios")):
        p = f"decoder.block.{i}"
        x = tap(f"{p}.conv_t1", F.conv_transpose1d(snake(x, W[p + ".snake1.alpha"]), W[p + ".conv_t1.weight"], W[p + ".conv_t1.bias"],
                                                   stride=s, padding=math.ceil(s / 2)))
        for u, d in enumerate(_cfg_get(cfg, "dilations"), start=1):
            x = tap(f"{p}.res_unit{u}", res_unit(x, W, f"{p}.res_unit{u}", d))
    x = F.conv1d(snake(x, W["decoder.snake1.alpha"]), W["decoder.conv2.weight"], W["decoder.conv2.bias"], padding=3)
    return tap("decoder.conv2", torch.tanh(x))


# --------------------------------------------------------------------------------------------
# residual vector quantiser with factorised, L2-normalised codes
# --------------------------------------------------------------------------------------------


def vq_stage(W, q: int, residual, variant: str = "descript", return_margin: bool = False):
    """dac/nn/quantize.py VectorQuantize.forward ([HF]:123-172) in eval mode on residual [B,H,N]:
        z_e = in_proj(residual)                         1x1 conv H -> D (with bias)
        idx = nearest code of normalize(z_e) among normalize(codebook) (search below)
        z_q = codebook[idx]  (UN-normalised);   z_q = z_e + (z_q - z_e)   (straight-through form, kept: it rounds)
        return out_proj(z_q)                            1x1 conv D -> H (with bias)
    """
    p = f"quantizer.quantizers.{q}"
    z_e = F.conv1d(residual, W[p + ".in_proj.weight"], W[p + ".in_proj.bias"])
    B, D, N = z_e.shape
    enc = z_e.permute(0, 2, 1).reshape(B * N, D)
    cb = W[p + ".codebook.weight"]
    e_n, c_n = F.normalize(enc), F.normalize(cb)
    if variant == "descript":
        dist = e_n.pow(2).sum(1, keepdim=True) - 2 * e_n @ c_n.t() + c_n.pow(2).sum(1, keepdim=True).t()
        score = -dist
    else:  # "hf": [HF]:167-168
        score = -(e_n.pow(2).sum(1, keepdim=True) - 2 * e_n @ c_n.t()) + c_n.pow(2).sum(1, keepdim=True).t()
    idx = score.max(1)[1]
    z_q = F.embedding(idx.view(B, N), cb).transpose(1, 2)
    z_q = z_e + (z_q - z_e)
    out = F.conv1d(z_q, W[p + ".out_proj.weight"], W[p + ".out_proj.bias"])
    if not return_margin:
        return out, idx.view(B, N), z_e
    d = 2.0 - 2.0 * (e_n @ c_n.t())                      # distance between unit vectors, for the tie audit
    two = torch.topk(d, 2, dim=1, largest=False).values
    margin = ((two[:, 1] - two[:, 0]) / two[:, 1].clamp_min(1e-30)).view(B, N)
    return out, idx.view(B, N), z_e, margin


def rvq_forward(cfg, W, z, K: int, variant: str = "descript", return_margin: bool = False):
    """dac/nn/quantize.py ResidualVectorQuantize.forward ([HF]:283-345), eval: z_q = sum_i z_q_i,
    residual -= z_q_i; stops after K quantisers.  -> z_q [B,H,N], codes [B,K,N] (, margins [B,K,N])."""
    z_q = 0
    residual = z
    codes, margins = [], []
    for i in range(min(K, _cfg_get(cfg, "n_codebooks"))):
        res = vq_stage(W, i, residual, variant, return_margin)
        z_q = z_q + res[0]
        residual = residual - res[0]
        codes.append(res[1])
        if return_margin:
            margins.append(res[3])
    codes = torch.stack(codes, dim=1)
    return (z_q, codes, torch.stack(margins, dim=1)) if return_margin else (z_q, codes)


def from_codes(cfg, W, codes):
    """dac/nn/quantize.py ResidualVectorQuantize.from_codes ([HF]:347-371): z_q = 0.0 + sum_i
    out_proj_i(codebook_i[codes[:, i]]) in stage order; codes [B,K,N] -> [B,H,N] (and latents [B,K*D,N])."""
    z_q = 0.0
    z_p = []
    for i in range(codes.shape[1]):
        p = f"quantizer.quantizers.{i}"
        z_p_i = F.embedding(codes[:, i], W[p + ".codebook.weight"]).transpose(1, 2)
        z_p.append(z_p_i)
        z_q = z_q + F.conv1d(z_p_i, W[p + ".out_proj.weight"], W[p + ".out_proj.bias"])
    return z_q, torch.cat(z_p, dim=1)


# --------------------------------------------------------------------------------------------
# the wrapper entry points
# --------------------------------------------------------------------------------------------


def _dtype(W):
    return W["encoder.conv1.bias"].dtype


def sig_to_toks(cfg, W, sig, length=None, num_codebooks: int = 8, variant: str = "descript", return_margin: bool = False):
    """dac.py:93-100: model.encode(sig[:,None], n_quantizers=K) -> codes [B,K,N] -> movedim -> [B,N,K].
    No padding to a hop multiple happens on this route (the wrapper never calls `preprocess`)."""
    z = encoder(cfg, W, sig[:, None].to(_dtype(W)))
    res = rvq_forward(cfg, W, z, num_codebooks, variant, return_margin)
    toks = res[1].movedim(-1, -2).contiguous()
    return (toks, res[2].movedim(-1, -2).contiguous()) if return_margin else toks


def sig_to_feats(cfg, W, sig, length=None, latent: bool = False, taps=None):
    """dac.py:103-112: encoder output [B,N,H]; with latent=True pushed through quantizers[0].in_proj -> [B,N,D]."""
    z = encoder(cfg, W, sig[:, None].to(_dtype(W)), taps)
    if latent:
        z = F.conv1d(z, W["quantizer.quantizers.0.in_proj.weight"], W["quantizer.quantizers.0.in_proj.bias"])
    return z.movedim(-1, -2)


def sig_to_qfeats(cfg, W, sig, length=None, num_codebooks: int = 8, variant: str = "descript"):
    """dac.py:115-120: the quantised representation returned by model.encode -> [B,N,H]."""
    z = encoder(cfg, W, sig[:, None].to(_dtype(W)))
    return rvq_forward(cfg, W, z, num_codebooks, variant)[0].movedim(-1, -2)


def toks_to_sig(cfg, W, toks, taps=None):
    """dac.py:123-130: quantizer.from_codes(toks.movedim(-1,-2)) -> model.decode(z_q)[:, 0] -> [B,T']."""
    z_q, _ = from_codes(cfg, W, toks.movedim(-1, -2))
    if taps is not None:
        taps["quantizer.from_codes"] = z_q
    return decoder(cfg, W, z_q, taps)[:, 0]


def embs(cfg, W, num_codebooks: int, latent: bool = False):
    """dac.py:63-90: latent=True -> stacked codebooks [K,C,D]; else every code through its quantiser's
    out_proj (with bias) -> [K,C,H]."""
    cbs = [W[f"quantizer.quantizers.{i}.codebook.weight"] for i in range(num_codebooks)]
    if latent:
        return torch.stack(cbs)
    out = []
    for i, cb in enumerate(cbs):
        p = f"quantizer.quantizers.{i}.out_proj"
        out.append(F.conv1d(cb[:, :, None], W[p + ".weight"], W[p + ".bias"])[..., 0])
    return torch.stack(out)
