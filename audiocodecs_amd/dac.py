"""DAC on MI355X -- host-side mirror of the reference wrapper `audiocodecs.DAC`
(/root/reference/audiocodecs/dac.py:28-130): same constructor arguments (`sample_rate`,
`orig_sample_rate=16000`, `mode`, `num_codebooks=8`, `latent=False`), attributes, method names and tensor
layouts.  The reference's backend `dac.DAC` (descript-audio-codec 1.0.0, dac.py:44,56-57) is replaced by the
gfx950 kernels behind the C ABI (include/audiocodecs_amd.h, ac_dac_create).

PARITY NOTE: descript-audio-codec is not installed in the build container, so this path is pinned to the
same-architecture `transformers.DacModel` only (tests/golden/dac_golden.npz); parity with the reference's
own backend is unpinned (oracle/dac_oracle.py header).
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _native
from .codec import Codec
from .config import DAC_16KHZ, DAC_24KHZ, DAC_44KHZ, DacConfig
from .encodec import _ptr, _stream

__all__ = ["DAC", "state_dict_from_descript"]

_BY_TAG = {16: DAC_16KHZ, 24: DAC_24KHZ, 44: DAC_44KHZ}


def state_dict_from_descript(sd: Dict[str, torch.Tensor], cfg: DacConfig) -> Dict[str, torch.Tensor]:
    """descript-audio-codec `DAC.state_dict()` (nn.Sequential indices, old-style weight-norm `weight_g`/
    `weight_v`) -> the HF-style names this library loads, with weight-norm folded (w = g * v / |v|, norm over
    dims 1,2 per index of dim 0).  Written from the published module structure (dac/model/dac.py); it could
    not be exercised against a real checkpoint offline."""
    nb = len(cfg.downsampling_ratios)
    names = {"encoder.block.0": "encoder.conv1", f"encoder.block.{nb + 1}": "encoder.snake1", f"encoder.block.{nb + 2}": "encoder.conv2",
             "decoder.model.0": "decoder.conv1", f"decoder.model.{nb + 1}": "decoder.snake1", f"decoder.model.{nb + 2}": "decoder.conv2"}
    unit = {0: "snake1", 1: "conv1", 2: "snake2", 3: "conv2"}
    for i in range(nb):
        for u in range(len(cfg.dilations)):
            for j, nm in unit.items():
                names[f"encoder.block.{i + 1}.block.{u}.block.{j}"] = f"encoder.block.{i}.res_unit{u + 1}.{nm}"
                names[f"decoder.model.{i + 1}.block.{u + 2}.block.{j}"] = f"decoder.block.{i}.res_unit{u + 1}.{nm}"
        nu = len(cfg.dilations)
        names[f"encoder.block.{i + 1}.block.{nu}"] = f"encoder.block.{i}.snake1"
        names[f"encoder.block.{i + 1}.block.{nu + 1}"] = f"encoder.block.{i}.conv1"
        names[f"decoder.model.{i + 1}.block.0"] = f"decoder.block.{i}.snake1"
        names[f"decoder.model.{i + 1}.block.1"] = f"decoder.block.{i}.conv_t1"
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        prefix, leaf = k.rsplit(".", 1)
        new = names.get(prefix, prefix)  # quantizer.* keeps its names
        if leaf == "weight_g":
            vv = sd[prefix + ".weight_v"].float()
            out[new + ".weight"] = vv * (v.float() / vv.flatten(1).norm(dim=1).view(-1, *([1] * (vv.dim() - 1))))
        elif leaf == "weight_v":
            continue
        else:
            out[f"{new}.{leaf}"] = v
    return out


class _NativeDac:
    """One DAC ac_handle: weights on one GPU + a grow-only workspace tensor."""

    def __init__(self, cfg: DacConfig, sd: Dict[str, torch.Tensor], device: torch.device, precision=None):
        self.lib = _native.lib()
        c = _native.AcDacConfig()
        c.struct_size = C.sizeof(_native.AcDacConfig)
        c.sampling_rate = cfg.sampling_rate
        c.encoder_hidden_size = cfg.encoder_hidden_size
        c.decoder_hidden_size = cfg.decoder_hidden_size
        c.num_ratios = len(cfg.downsampling_ratios)
        for i, r in enumerate(cfg.downsampling_ratios):
            c.downsampling_ratios[i] = r
        for i, r in enumerate(cfg.upsampling_ratios):
            c.upsampling_ratios[i] = r
        c.n_codebooks = cfg.n_codebooks
        c.codebook_size = cfg.codebook_size
        c.codebook_dim = cfg.codebook_dim
        c.num_dilations = len(cfg.dilations)
        for i, d in enumerate(cfg.dilations):
            c.dilations[i] = d
        c.device = device.index if device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", c.device)
        self.h = C.c_void_p()
        rc = self.lib.ac_dac_create(C.byref(c), C.byref(self.h))
        if rc < 0:
            raise _native.NativeError(f"ac_dac_create failed with code {rc} (unsupported configuration, or no gfx950 GPU visible)")
        _native.set_precision(self.lib, self.h, precision)
        for name, t in sd.items():
            if not t.is_floating_point():
                continue
            t = t.detach().to(torch.float32).cpu().contiguous()
            _native.check(
                self.lib.ac_load_weights(self.h, name.encode(), C.c_void_p(t.data_ptr()), t.numel() * 4),
                self.h, f"ac_load_weights({name})",
            )
        with torch.cuda.device(self.device):
            _native.check(self.lib.ac_finalize(self.h), self.h, "ac_finalize")
        self.ws: Optional[torch.Tensor] = None
        _native.track(self)

    def workspace(self, nbytes: int) -> torch.Tensor:
        if self.ws is None or self.ws.numel() < nbytes:
            self.ws = None
            self.ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return self.ws

    def __del__(self):
        try:
            import sys

            if sys.is_finalizing():   # interpreter shutdown: the HIP runtime may already be gone, the OS reclaims the rest
                return
            if getattr(self, "h", None):
                self.lib.ac_destroy(self.h)
                self.h = None
        except Exception:
            pass


class DAC(Codec):
    _accepts_none_length = True

    def __init__(
        self,
        sample_rate,
        orig_sample_rate=16000,
        mode="reconstruct",
        num_codebooks=8,
        latent=False,
        *,
        state_dict: Optional[Dict[str, torch.Tensor]] = None,
        config: Optional[DacConfig] = None,
        precision: Optional[str] = None,
        strict: bool = False,
        graph: bool = False,
    ):
        """`state_dict`: HF `DacModel.state_dict()` names, or descript's own (`weights.pth["state_dict"]`, detected
        by its `weight_g` keys and converted by :func:`state_dict_from_descript`), or
        `checkpoint.synthetic_dac_state_dict(cfg, seed)`.  Without it the reference downloads the checkpoint
        through `dac.utils.download` (dac.py:56) -- that package is not a dependency here, so it must be given."""
        super().__init__(sample_rate, orig_sample_rate, mode)
        self.strict = bool(strict)   # codec.py: poll the handle after every call
        self.graph = bool(graph)     # codec.py: replay one hipGraph per (call, shape)
        self.num_codebooks = num_codebooks
        self.vocab_size = 1024  # dac.py:52
        self.latent = latent
        self.precision = _native.check_precision(precision)   # see Encodec: None / "fp32" (parity arithmetic), "fp32_exact"
        tag = int(orig_sample_rate / 1000)  # dac.py:55
        if config is None:
            if tag not in _BY_TAG:
                raise ValueError(f"no DAC model for {orig_sample_rate} Hz (16, 24 and 44.1 kHz exist)")
            config = _BY_TAG[tag]
        self.config = config
        if state_dict is None:
            raise ImportError("pass state_dict=: the pretrained DAC weights ship with `descript-audio-codec`, which is not a dependency")
        if any(k.endswith("weight_g") for k in state_dict):
            state_dict = state_dict_from_descript(state_dict, config)
        self._sd = dict(state_dict)
        self._natives: Dict[int, _NativeDac] = {}

    def _native_for(self, t: torch.Tensor) -> _NativeDac:
        if not t.is_cuda:
            raise _native.NativeError(
                "audiocodecs_amd runs on MI355X only: move the input to a cuda device (there is deliberately no CPU fallback)"
            )
        idx = t.device.index
        if idx not in self._natives:
            self._natives[idx] = _NativeDac(self.config, self._sd, t.device, self.precision)
        return self._natives[idx]

    def _any_native(self) -> _NativeDac:
        dev = next(iter(self._natives.values())).device if self._natives else torch.device("cuda", torch.cuda.current_device())
        return self._native_for(torch.empty(0, device=dev))

    def _K(self) -> int:
        # dac/nn/quantize.py: the loop breaks at i >= n_quantizers, so asking for more than exist uses them all
        if self.num_codebooks < 1:
            raise ValueError(f"num_codebooks must be >= 1, got {self.num_codebooks}")
        return min(self.num_codebooks, self.config.n_codebooks)

    def _frames(self, T: int) -> int:
        N = self.config.num_frames(T) if T >= 1 else 0
        if N < 1:
            raise RuntimeError(
                f"Kernel size can't be greater than actual input size: {T} samples are too short for the "
                f"strided convolutions (hop {self.config.hop_length})"
            )
        return N

    # override
    @torch.no_grad()
    def embs(self):
        nat = self._any_native()
        K = self._K()
        width = self.config.codebook_dim if self.latent else self.config.hidden_size
        out = torch.empty(K, self.vocab_size, width, device=nat.device)
        with torch.cuda.device(nat.device):
            fn = nat.lib.ac_embs if self.latent else nat.lib.ac_embs_projected
            _native.check(fn(nat.h, K, _ptr(out), _stream()), nat.h, "ac_embs")
        return out  # [K, C, 8] (latent) or [K, C, H]

    def _encode(self, sig, want_qfeats: bool):
        B, T = sig.shape
        K, N = self._K(), self._frames(T)
        toks = torch.empty(B, N, K, dtype=torch.int64, device=sig.device)
        qf = torch.empty(B, N, self.config.hidden_size, dtype=torch.float32, device=sig.device) if want_qfeats else None
        if B == 0:   # an empty shard (sharding.shard_bounds): nothing to run, the library is not called
            return toks, qf
        nat = self._native_for(sig)
        sig = sig.to(torch.float32).contiguous()
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_encode_workspace_bytes(nat.h, B, T))
            if want_qfeats:
                rc = nat.lib.ac_encode_quantized(nat.h, _ptr(sig), B, T, K, _ptr(toks), _ptr(qf), _ptr(ws), ws.numel(), _stream())
            else:
                rc = nat.lib.ac_encode(nat.h, _ptr(sig), None, B, T, K, _ptr(toks), _ptr(ws), ws.numel(), _stream())
            _native.check(rc, nat.h, "ac_encode")
        return toks, qf

    # override
    def _sig_to_toks(self, sig, length):
        # sig: [B, T] -> [B, N, K]; `length` is unused upstream (dac.py:93-100)
        return self._encode(sig, False)[0]

    # override
    def _sig_to_feats(self, sig, length):
        # sig: [B, T] -> encoder output [B, N, H], or quantizers[0].in_proj of it [B, N, 8] when latent (dac.py:103-112)
        B, T = sig.shape
        N = self._frames(T)
        width = self.config.codebook_dim if self.latent else self.config.hidden_size
        feats = torch.empty(B, N, width, dtype=torch.float32, device=sig.device)
        if B == 0:
            return feats
        nat = self._native_for(sig)
        sig = sig.to(torch.float32).contiguous()
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_encode_workspace_bytes(nat.h, B, T))
            if self.latent:
                rc = nat.lib.ac_encode_feats_latent(nat.h, _ptr(sig), B, T, _ptr(feats), _ptr(ws), ws.numel(), _stream())
            else:
                rc = nat.lib.ac_encode_feats(nat.h, _ptr(sig), None, B, T, _ptr(feats), _ptr(ws), ws.numel(), _stream())
            _native.check(rc, nat.h, "ac_encode_feats")
        return feats

    # override
    def _sig_to_qfeats(self, sig, length):
        # the quantised representation model.encode returns (dac.py:115-120) -> [B, N, H]
        return self._encode(sig, True)[1]

    # override
    def _toks_to_sig(self, toks, length):
        # toks: [B, N, K] -> quantizer.from_codes -> decoder -> [B, T'] (dac.py:123-130)
        B, N, K = toks.shape
        L = self.config.num_samples(N)
        sig = torch.empty(B, L, dtype=torch.float32, device=toks.device)
        if B == 0:
            return sig
        nat = self._native_for(toks)
        toks = toks.to(torch.int64).contiguous()
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_decode_workspace_bytes(nat.h, B, N))
            _native.check(
                nat.lib.ac_decode(nat.h, _ptr(toks), B, N, K, _ptr(sig), _ptr(ws), ws.numel(), _stream()),
                nat.h, "ac_decode",
            )
        return sig

    # ---- measurement hook used by bench.py ------------------------------------------------------
    def profile_kernels(self, fn):
        nat = self._any_native()
        _native.check(nat.lib.ac_profile_begin(nat.h), nat.h, "ac_profile_begin")
        try:
            fn()
        finally:
            buf = (_native.AcKernelStat * 256)()
            n = nat.lib.ac_profile_end(nat.h, buf, 256)
        _native.check(n, nat.h, "ac_profile_end")
        return [(buf[i].name.decode(), buf[i].launches, buf[i].total_ms, buf[i].flops, buf[i].bytes) for i in range(n)]
