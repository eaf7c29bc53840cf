"""ORACLE -- test infrastructure only; never imported by the product path.

CPU restatement (torch functional ops, dtype-generic: float32 or float64) of the EnCodec-24k
encode/decode path the reference wrapper runs:

    audiocodecs.Codec.sig_to_toks  -> Encodec._sig_to_toks   /root/reference/audiocodecs/codec.py:57-66,
                                                              /root/reference/audiocodecs/encodec.py:82-94
    audiocodecs.Codec.toks_to_sig  -> Encodec._toks_to_sig   codec.py:90-100, encodec.py:130-141

The arithmetic itself lives in a third-party dependency that is NOT under /root/reference:
`transformers` (pinned 4.46.3 in downstream/environment.yml:253; 5.15.0 in this image), file
``models/encodec/modeling_encodec.py`` -- cited below as [HF]:line (5.15.0 line numbers).

PARITY PIN: this restatement is checked against the reference wrapper itself
(``audiocodecs.encodec.Encodec`` imported from /root/reference in the build container, with seeded
synthetic weights -- tools/make_golden.py) through the fixtures in tests/golden/; see
tests/test_oracle_golden.py.  The reference repo holds no golden vectors of its own.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# architecture bookkeeping (own derivation; mirrors [HF]:285-347 module lists)
# --------------------------------------------------------------------------------------------


def _cfg_get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


def fold_weight_norm(sd: Dict[str, torch.Tensor], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """w = g * v / ||v||_2 over dims (1,2) per index of dim 0 ([HF]:106-111,194-199 apply
    torch.nn.utils.parametrizations.weight_norm with its default dim=0).  Evaluated in fp32 with the
    same torch primitive the parametrisation calls, then cast to `dtype`."""
    out = {}
    s0, s1 = ".parametrizations.weight.original0", ".parametrizations.weight.original1"
    for k, v in sd.items():
        if k.endswith(s0):
            p = k[: -len(s0)]
            out[p + ".weight"] = torch._weight_norm(sd[p + s1].float(), v.float(), 0).to(dtype)
        elif k.endswith(s1):
            continue
        else:
            out[k] = v.to(dtype) if v.is_floating_point() else v
    return out


# --------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------


def pad1d_reflect(x: torch.Tensor, left: int, right: int) -> torch.Tensor:
    """[HF]:139-155 `_pad1d` in reflect mode, including the small-input workaround: when the input
    is not longer than the largest pad, zero-extend on the right, reflect, drop the extension."""
    length = x.shape[-1]
    max_pad = max(left, right)
    extra = 0
    if length <= max_pad:
        extra = max_pad - length + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (left, right), mode="reflect")
    return y[..., : y.shape[-1] - extra]


def conv1d_causal(x, w, b, stride: int = 1, dilation: int = 1):
    """[HF]:157-176 EncodecConv1d.forward, causal branch, pad_mode='reflect'.
    x [B,Cin,L] -> [B,Cout,ceil(L/stride)]."""
    k_eff = (w.shape[-1] - 1) * dilation + 1
    pad_total = k_eff - stride
    length = x.shape[-1]
    # [HF]:126-136 _get_extra_padding_for_conv1d
    n_frames = math.ceil((length - k_eff + pad_total) / stride + 1) - 1
    extra = n_frames * stride + k_eff - pad_total - length
    x = pad1d_reflect(x, pad_total, extra)
    return F.conv1d(x, w, b, stride=stride, dilation=dilation)


def convtr1d_causal(x, w, b, stride: int):
    """[HF]:206-233 EncodecConvTranspose1d.forward, causal, trim_right_ratio=1.0: the whole fixed
    padding k - stride is trimmed on the right.  x [B,Cin,L] -> [B,Cout,L*stride]."""
    k = w.shape[-1]
    y = F.conv_transpose1d(x, w, b, stride=stride)
    pad_right = math.ceil((k - stride) * 1.0)
    pad_left = (k - stride) - pad_right
    return y[..., pad_left : y.shape[-1] - pad_right]


def resblock(x, W, p: str):
    """[HF]:277-282: shortcut(x) + conv_k1(ELU(conv_k3(ELU(x)))) (dilations (1,1), conv shortcut)."""
    h = conv1d_causal(F.elu(x), W[p + ".block.1.conv.weight"], W[p + ".block.1.conv.bias"])
    h = conv1d_causal(F.elu(h), W[p + ".block.3.conv.weight"], W[p + ".block.3.conv.bias"])
    return conv1d_causal(x, W[p + ".shortcut.conv.weight"], W[p + ".shortcut.conv.bias"]) + h


def lstm_skip(x, W, p: str, num_layers: int, explicit: bool = False):
    """[HF]:245-249 EncodecLSTM: [B,C,T] -> [T,B,C]; torch LSTM, zero initial state; out + in; back.
    `explicit=True` spells the recurrence out (gate order i,f,g,o); the default calls ATen's lstm,
    the kernel nn.LSTM itself dispatches to."""
    xt = x.permute(2, 0, 1)
    T, B, C = xt.shape
    if not explicit:
        flat = []
        for l in range(num_layers):
            flat += [W[f"{p}.weight_ih_l{l}"], W[f"{p}.weight_hh_l{l}"], W[f"{p}.bias_ih_l{l}"], W[f"{p}.bias_hh_l{l}"]]
        h0 = xt.new_zeros(num_layers, B, C)
        y = torch._VF.lstm(xt.contiguous(), (h0, h0.clone()), flat, True, num_layers, 0.0, False, False, False)[0]
    else:
        y = xt
        for l in range(num_layers):
            w_ih, w_hh = W[f"{p}.weight_ih_l{l}"], W[f"{p}.weight_hh_l{l}"]
            b_ih, b_hh = W[f"{p}.bias_ih_l{l}"], W[f"{p}.bias_hh_l{l}"]
            h = xt.new_zeros(B, C)
            c = xt.new_zeros(B, C)
            outs = []
            for t in range(T):
                g = y[t] @ w_ih.T + b_ih + h @ w_hh.T + b_hh
                i, f, gg, o = g.chunk(4, dim=-1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
                outs.append(h)
            y = torch.stack(outs)
    return (y + xt).permute(1, 2, 0)


def encoder(cfg, W, x, taps: Optional[dict] = None, explicit_lstm: bool = False):
    """[HF]:285-313 EncodecEncoder.  x [B,1,T] -> [B,hidden,N]."""
    ratios = list(_cfg_get(cfg, "upsampling_ratios"))
    p = "encoder.layers."
    h = conv1d_causal(x, W[p + "0.conv.weight"], W[p + "0.conv.bias"])
    if taps is not None:
        taps["enc0"] = h
    i = 1
    for r in reversed(ratios):
        h = resblock(h, W, f"{p}{i}")
        if taps is not None:
            taps[f"enc{i}"] = h
        h = conv1d_causal(F.elu(h), W[f"{p}{i + 2}.conv.weight"], W[f"{p}{i + 2}.conv.bias"], stride=r)
        if taps is not None:
            taps[f"enc{i + 2}"] = h
        i += 3
    h = lstm_skip(h, W, f"{p}{i}.lstm", _cfg_get(cfg, "num_lstm_layers"), explicit_lstm)
    if taps is not None:
        taps[f"enc{i}"] = h
    h = conv1d_causal(F.elu(h), W[f"{p}{i + 2}.conv.weight"], W[f"{p}{i + 2}.conv.bias"])
    return h


def decoder(cfg, W, z, taps: Optional[dict] = None, explicit_lstm: bool = False):
    """[HF]:316-347 EncodecDecoder.  z [B,hidden,N] -> [B,1,N*hop]."""
    ratios = list(_cfg_get(cfg, "upsampling_ratios"))
    p = "decoder.layers."
    h = conv1d_causal(z, W[p + "0.conv.weight"], W[p + "0.conv.bias"])
    if taps is not None:
        taps["dec0"] = h
    h = lstm_skip(h, W, p + "1.lstm", _cfg_get(cfg, "num_lstm_layers"), explicit_lstm)
    if taps is not None:
        taps["dec1"] = h
    i = 2
    for r in ratios:
        h = convtr1d_causal(F.elu(h), W[f"{p}{i + 1}.conv.weight"], W[f"{p}{i + 1}.conv.bias"], r)
        if taps is not None:
            taps[f"dec{i + 1}"] = h
        h = resblock(h, W, f"{p}{i + 2}")
        if taps is not None:
            taps[f"dec{i + 2}"] = h
        i += 3
    h = conv1d_causal(F.elu(h), W[f"{p}{i + 1}.conv.weight"], W[f"{p}{i + 1}.conv.bias"])
    return h


def codebooks(W, K: int) -> List[torch.Tensor]:
    return [W[f"quantizer.layers.{k}.codebook.embed"] for k in range(K)]


def rvq_encode(embs: Sequence[torch.Tensor], z, return_margin: bool = False):
    """[HF]:424-438 RVQ.encode over [HF]:364-369 EuclideanCodebook.quantize.

    z [B,H,N] -> codes [K,B,N] int64.  Per stage, frames flattened (b,n)-major:
        dist = -(sum(x^2) - 2 x @ E^T + sum(E^2));  idx = dist.max(-1).indices (first max on ties);
        residual <- residual - E[idx].
    With return_margin also returns, per token, (best - second_best) / max(|best|, tiny): the
    relative gap used by the near-tie policy of the parity tests."""
    residual = z
    B, H, N = z.shape
    out, margins = [], []
    for E in embs:
        x = residual.permute(0, 2, 1).reshape(-1, H)
        et = E.t()
        dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ et + et.pow(2).sum(0, keepdim=True))
        idx = dist.max(dim=-1).indices
        if return_margin:
            top2 = dist.topk(2, dim=-1).values
            margins.append(((top2[:, 0] - top2[:, 1]) / top2[:, 0].abs().clamp_min(1e-30)).view(B, N))
        q = F.embedding(idx.view(B, N), E).permute(0, 2, 1)
        residual = residual - q
        out.append(idx.view(B, N))
    codes = torch.stack(out)
    if return_margin:
        return codes, torch.stack(margins)
    return codes


def rvq_decode(embs: Sequence[torch.Tensor], codes):
    """[HF]:440-447: sum_k embedding(codes[k], E_k), accumulated in k order from scalar 0.0.
    codes [K,B,N] -> [B,H,N]."""
    q_out = torch.zeros((), dtype=embs[0].dtype)
    for k, idx in enumerate(codes):
        q_out = q_out + F.embedding(idx, embs[k]).permute(0, 2, 1)
    return q_out


# --------------------------------------------------------------------------------------------
# wrapper-level entry points (the drop-in boundary)
# --------------------------------------------------------------------------------------------


def num_quantizers_for(cfg, num_codebooks: int) -> int:
    """encodec.py:50 bandwidth = K*75/100, then [HF]:416-422 get_num_quantizers_for_bandwidth."""
    bandwidth = (num_codebooks * 75) / 100
    tb = tuple(_cfg_get(cfg, "target_bandwidths"))
    if bandwidth not in tb:
        raise ValueError(f"This model doesn't support the bandwidth {bandwidth}. Select one of {list(tb)}.")
    hop = math.prod(_cfg_get(cfg, "upsampling_ratios"))
    frame_rate = math.ceil(_cfg_get(cfg, "sampling_rate") / hop)
    bw_per_q = math.log2(_cfg_get(cfg, "codebook_size")) * frame_rate
    return int(max(1, math.floor(bandwidth * 1000 / bw_per_q)))


def padding_mask(sig, length):
    """encodec.py:84-89: abs_lens = T*length; mask = arange(max_len) < abs_lens."""
    abs_lens = sig.shape[-1] * length
    max_len = abs_lens.max().long().item()
    return torch.arange(max_len, dtype=length.dtype)[None] < abs_lens[:, None]


def masked_embeddings(cfg, W, sig, length=None, taps=None, explicit_lstm=False):
    """Masked input -> encoder output [B,H,N] (the embeddings RVQ sees).  encodec.py:82-92 and
    [HF]:589-590 (frame = mask * input_values; normalize=False for the 24 kHz model)."""
    if length is None:
        length = torch.ones(len(sig))  # codec.py:64-65
    mask = padding_mask(sig, length)
    x = (mask * sig)[:, None].to(W["encoder.layers.0.conv.bias"].dtype)
    return encoder(cfg, W, x, taps, explicit_lstm)


def sig_to_feats(cfg, W, sig, length=None, taps=None, explicit_lstm=False):
    """encodec.py:97-118 `_sig_to_feats` -> [B,N,H].  NOTE the reference applies the padding mask
    here only `if self.model.config.normalize` (:107-112) -- False for the 24 kHz model -- so the
    encoder sees the UNMASKED signal; `length` has no effect on this entry point."""
    x = sig[:, None].to(W["encoder.layers.0.conv.bias"].dtype)
    return encoder(cfg, W, x, taps, explicit_lstm).movedim(-1, -2)


def sig_to_toks(cfg, W, sig, length=None, num_codebooks: int = 8, return_margin: bool = False):
    """sig [B,T] -> toks [B,N,K] int64 (encodec.py:93 movedim(-1,-2) of codes.transpose(0,1))."""
    K = num_quantizers_for(cfg, num_codebooks)
    z = masked_embeddings(cfg, W, sig, length)
    res = rvq_encode(codebooks(W, K), z, return_margin)
    if return_margin:
        codes, m = res
        return codes.permute(1, 2, 0).contiguous(), m.permute(1, 2, 0).contiguous()
    return res.permute(1, 2, 0).contiguous()


def sig_to_qfeats(cfg, W, sig, length=None, num_codebooks: int = 8):
    """encodec.py:121-127: toks -> quantizer.decode -> [B,N,H]."""
    return toks_to_qfeats(cfg, W, sig_to_toks(cfg, W, sig, length, num_codebooks))


def toks_to_qfeats(cfg, W, toks):
    """encodec.py:144-149: quantizer.decode(toks.movedim(-1,0)).movedim(-1,-2) -> [B,N,H]."""
    codes = toks.movedim(-1, 0)
    return rvq_decode(codebooks(W, codes.shape[0]), codes).movedim(-1, -2)


def toks_to_sig(cfg, W, toks, taps=None, explicit_lstm=False):
    """toks [B,N,K] -> sig [B, N*hop] (encodec.py:139-140; output is NOT trimmed to the input
    length: [HF]:705-707 only truncates when a padding_mask is given, the wrapper passes none)."""
    codes = toks.movedim(-1, 0)
    z = rvq_decode(codebooks(W, codes.shape[0]), codes)
    return decoder(cfg, W, z, taps, explicit_lstm)[:, 0]


def embs(W, num_codebooks: int):
    """encodec.py:74-79: stacked codebooks [K,C,H]."""
    return torch.stack(codebooks(W, num_codebooks))
