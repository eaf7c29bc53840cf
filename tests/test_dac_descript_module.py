"""descript-audio-codec's checkpoint layout, produced by TORCH ITSELF: a module tree built the way the published
dac/model/dac.py, dac/nn/layers.py and dac/nn/quantize.py build theirs (nn.Sequential nesting, old-style
torch.nn.utils.weight_norm -> `weight_g` / `weight_v`, Snake1d `alpha`), whose `state_dict()` keys are whatever torch names
them -- not a renaming table written next to the mapping under test (tests/test_dac_state_dict.py is that self-check).
`audiocodecs_amd.dac.state_dict_from_descript` must turn those keys into the loader's names such that
  * the CPU oracle on the mapped weights and the module tree's own forward (encoder -> residual VQ -> decoder) agree:
    identical token ids, latents and waveform within fp32 rounding -- which also checks WHICH Sequential index is which layer
    (a swapped snake / conv index would still load, but compute something else);
  * every tensor the loader expects is present.
This is the reference's load path (/root/reference/audiocodecs/dac.py:56-57: dac.utils.download + dac.DAC.load) restated from
the published source; the real package is not on disk, so the row stays "parity unpinned"."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from audiocodecs_amd import checkpoint
from audiocodecs_amd.config import DAC_TINY
from audiocodecs_amd.dac import state_dict_from_descript
from oracle import dac_oracle as O


def WNConv1d(*a, **k):
    return torch.nn.utils.weight_norm(nn.Conv1d(*a, **k))


def WNConvTranspose1d(*a, **k):
    return torch.nn.utils.weight_norm(nn.ConvTranspose1d(*a, **k))


class Snake1d(nn.Module):      # dac/nn/layers.py: x + (alpha + 1e-9)^-1 sin^2(alpha x), alpha [1, C, 1]
    def __init__(self, channels):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, channels, 1))

    def forward(self, x):
        return x + (self.alpha + 1e-9).reciprocal() * torch.sin(self.alpha * x).pow(2)


class ResidualUnit(nn.Module):
    def __init__(self, dim, dilation):
        super().__init__()
        pad = ((7 - 1) * dilation) // 2
        self.block = nn.Sequential(Snake1d(dim), WNConv1d(dim, dim, kernel_size=7, dilation=dilation, padding=pad), Snake1d(dim),
                                   WNConv1d(dim, dim, kernel_size=1))

    def forward(self, x):
        y = self.block(x)
        pad = (x.shape[-1] - y.shape[-1]) // 2
        if pad > 0:
            x = x[..., pad:-pad]
        return x + y


class EncoderBlock(nn.Module):
    def __init__(self, dim, stride, dilations):
        super().__init__()
        self.block = nn.Sequential(*[ResidualUnit(dim // 2, d) for d in dilations], Snake1d(dim // 2),
                                   WNConv1d(dim // 2, dim, kernel_size=2 * stride, stride=stride, padding=math.ceil(stride / 2)))

    def forward(self, x):
        return self.block(x)


class Encoder(nn.Module):
    def __init__(self, d_model, strides, d_latent, dilations):
        super().__init__()
        blocks = [WNConv1d(1, d_model, kernel_size=7, padding=3)]
        for s in strides:
            d_model *= 2
            blocks.append(EncoderBlock(d_model, s, dilations))
        blocks += [Snake1d(d_model), WNConv1d(d_model, d_latent, kernel_size=3, padding=1)]
        self.block = nn.Sequential(*blocks)

    def forward(self, x):
        return self.block(x)


class DecoderBlock(nn.Module):
    def __init__(self, din, dout, stride, dilations):
        super().__init__()
        self.block = nn.Sequential(Snake1d(din), WNConvTranspose1d(din, dout, kernel_size=2 * stride, stride=stride, padding=math.ceil(stride / 2)),
                                   *[ResidualUnit(dout, d) for d in dilations])

    def forward(self, x):
        return self.block(x)


class Decoder(nn.Module):
    def __init__(self, d_latent, channels, rates, dilations):
        super().__init__()
        layers = [WNConv1d(d_latent, channels, kernel_size=7, padding=3)]
        for i, s in enumerate(rates):
            layers.append(DecoderBlock(channels // 2 ** i, channels // 2 ** (i + 1), s, dilations))
        out = channels // 2 ** len(rates)
        layers += [Snake1d(out), WNConv1d(out, 1, kernel_size=7, padding=3), nn.Tanh()]
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class VectorQuantize(nn.Module):      # dac/nn/quantize.py (eval path)
    def __init__(self, input_dim, codebook_size, codebook_dim):
        super().__init__()
        self.in_proj = WNConv1d(input_dim, codebook_dim, kernel_size=1)
        self.out_proj = WNConv1d(codebook_dim, input_dim, kernel_size=1)
        self.codebook = nn.Embedding(codebook_size, codebook_dim)

    def forward(self, z):
        z_e = self.in_proj(z)
        enc = z_e.transpose(1, 2).reshape(-1, z_e.shape[1])
        cb = self.codebook.weight
        enc_n, cb_n = F.normalize(enc), F.normalize(cb)
        dist = enc_n.pow(2).sum(1, keepdim=True) - 2 * enc_n @ cb_n.t() + cb_n.pow(2).sum(1, keepdim=True).t()
        idx = (-dist).max(1)[1].view(z.shape[0], -1)
        z_q = F.embedding(idx, cb).transpose(1, 2)
        z_q = z_e + (z_q - z_e).detach()
        return self.out_proj(z_q), idx


class ResidualVectorQuantize(nn.Module):
    def __init__(self, input_dim, n_codebooks, codebook_size, codebook_dim):
        super().__init__()
        self.quantizers = nn.ModuleList([VectorQuantize(input_dim, codebook_size, codebook_dim) for _ in range(n_codebooks)])

    def forward(self, z, n):
        zq, res, codes = 0, z, []
        for q in self.quantizers[:n]:
            zi, idx = q(res)
            zq = zq + zi
            res = res - zi
            codes.append(idx)
        return zq, torch.stack(codes, 1)


class DescriptDAC(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        latent = cfg.encoder_hidden_size * 2 ** len(cfg.downsampling_ratios)
        self.encoder = Encoder(cfg.encoder_hidden_size, cfg.downsampling_ratios, latent, cfg.dilations)
        self.quantizer = ResidualVectorQuantize(latent, cfg.n_codebooks, cfg.codebook_size, cfg.codebook_dim)
        self.decoder = Decoder(latent, cfg.decoder_hidden_size, cfg.upsampling_ratios, cfg.dilations)


def _randomise(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("alpha"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif name.endswith("weight_g"):
                p.mul_(0.8 + 0.4 * torch.rand(p.shape, generator=g))      # g != |v|: the fold must really apply g / |v|
            elif name.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif "codebook" in name:
                p.copy_(torch.randn(p.shape, generator=g))


def test_torch_named_descript_checkpoint_maps_onto_the_loader_and_computes_the_same():
    cfg = DAC_TINY
    torch.manual_seed(0)
    model = DescriptDAC(cfg).eval()
    _randomise(model, 5)
    sd_descript = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert any(k.endswith(".weight_g") for k in sd_descript) and "encoder.block.0.weight_v" in sd_descript
    mapped = state_dict_from_descript(sd_descript, cfg)
    want_keys = set(checkpoint.synthetic_dac_state_dict(cfg, seed=0))
    assert set(mapped) == want_keys, (sorted(want_keys - set(mapped))[:5], sorted(set(mapped) - want_keys)[:5])
    W = O.cast_weights(mapped)
    g = torch.Generator().manual_seed(9)
    sig = 0.3 * torch.randn(2, 3200, generator=g)
    K = cfg.n_codebooks
    with torch.no_grad():
        z = model.encoder(sig[:, None])
        zq, codes = model.quantizer(z, K)
        rec = model.decoder(zq)[:, 0]
        oz = O.sig_to_feats(cfg, W, sig)                      # [B, N, H]
        otoks = O.sig_to_toks(cfg, W, sig, None, K)           # [B, N, K]
        orec = O.toks_to_sig(cfg, W, codes.movedim(1, 2))
    np.testing.assert_allclose(oz.numpy(), z.transpose(1, 2).numpy(), rtol=0, atol=2e-5 * float(z.abs().max()))
    assert torch.equal(otoks, codes.movedim(1, 2))
    assert orec.shape == rec.shape
    np.testing.assert_allclose(orec.numpy(), rec.numpy(), rtol=0, atol=2e-5)
