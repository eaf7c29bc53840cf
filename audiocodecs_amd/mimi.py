"""Mimi on MI355X -- host-side mirror of the reference wrapper `audiocodecs.Mimi`
(/root/reference/audiocodecs/mimi.py:25-156): same constructor arguments (`sample_rate`, `mode`,
`num_codebooks`, `latent`), attributes (`num_codebooks`, `vocab_size`, `latent`), method names, tensor
layouts and error behaviour.  The third-party `transformers.MimiModel` the reference calls
(mimi.py:45,105,115-119,139,146,153) is replaced by the gfx950 kernels behind the C ABI
(include/audiocodecs_amd.h, ac_mimi_create).  PyTorch is used for device memory and streams only.
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _native
from .codec import Codec
from .config import MIMI_24KHZ, MimiConfig
from .encodec import _ptr, _stream

__all__ = ["Mimi"]


class _NativeMimi:
    """One Mimi ac_handle: weights on one GPU + a grow-only workspace tensor."""

    def __init__(self, cfg: MimiConfig, sd: Dict[str, torch.Tensor], device: torch.device, precision=None):
        self.lib = _native.lib()
        c = _native.AcMimiConfig()
        c.struct_size = C.sizeof(_native.AcMimiConfig)
        for f in ("sampling_rate", "num_filters", "hidden_size", "kernel_size", "last_kernel_size", "residual_kernel_size",
                  "compress", "codebook_size", "codebook_dim", "num_quantizers", "num_semantic_quantizers", "num_hidden_layers",
                  "num_attention_heads", "head_dim", "intermediate_size", "sliding_window", "resample_stride"):
            setattr(c, f, getattr(cfg, f))
        c.num_ratios = len(cfg.upsampling_ratios)
        for i, r in enumerate(cfg.upsampling_ratios):
            c.upsampling_ratios[i] = r
        c.rope_theta = cfg.rope_theta
        c.norm_eps = cfg.norm_eps
        c.device = device.index if device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", c.device)
        self.h = C.c_void_p()
        rc = self.lib.ac_mimi_create(C.byref(c), C.byref(self.h))
        if rc < 0:
            raise _native.NativeError(f"ac_mimi_create failed with code {rc} (is a gfx950 GPU visible?)")
        _native.set_precision(self.lib, self.h, precision)
        for name, t in sd.items():
            if not t.is_floating_point() or name.endswith(".initialized"):
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


class Mimi(Codec):
    _accepts_none_length = True

    def __init__(
        self,
        sample_rate,
        mode="reconstruct",
        num_codebooks=8,
        latent=True,
        *,
        state_dict: Optional[Dict[str, torch.Tensor]] = None,
        config: MimiConfig = MIMI_24KHZ,
        precision: Optional[str] = None,
        strict: bool = False,
        graph: bool = False,
    ):
        """`state_dict`: an HF-format MimiModel state dict (`safetensors.torch.load_file` of kyutai/mimi's
        model.safetensors, or `checkpoint.synthetic_mimi_state_dict(cfg, seed)`); fetched through
        huggingface_hub like the reference when omitted (needs network or a warm cache)."""
        super().__init__(sample_rate, config.sampling_rate, mode)  # mimi.py:38
        self.strict = bool(strict)   # codec.py: poll the handle after every call
        self.graph = bool(graph)     # codec.py: replay one hipGraph per (call, shape)
        self.num_codebooks = num_codebooks
        self.vocab_size = config.codebook_size  # 2048 (mimi.py:40)
        self.latent = latent
        self.config = config
        self.precision = _native.check_precision(precision)   # see Encodec: None / "fp32" (parity arithmetic), "fp32_exact"
        if state_dict is None:
            state_dict = self._fetch_pretrained()
        self._sd = {k: v for k, v in state_dict.items()}
        self._natives: Dict[int, _NativeMimi] = {}

    @staticmethod
    def _fetch_pretrained():
        try:
            from huggingface_hub import hf_hub_download
            from safetensors.torch import load_file
        except ImportError:
            raise ImportError("`pip install huggingface_hub safetensors` to fetch pretrained Mimi weights")
        return load_file(hf_hub_download("kyutai/mimi", "model.safetensors"))

    # ------------------------------------------------------------------------------------------
    def _native_for(self, t: torch.Tensor) -> _NativeMimi:
        if not t.is_cuda:
            raise _native.NativeError(
                "audiocodecs_amd runs on MI355X only: move the input to a cuda device "
                "(there is deliberately no CPU fallback)"
            )
        idx = t.device.index
        if idx not in self._natives:
            self._natives[idx] = _NativeMimi(self.config, self._sd, t.device, self.precision)
        return self._natives[idx]

    def _any_native(self) -> _NativeMimi:
        dev = next(iter(self._natives.values())).device if self._natives else torch.device("cuda", torch.cuda.current_device())
        return self._native_for(torch.empty(0, device=dev))

    def _check_num_codebooks(self):
        """[HF] mimi :1106-1114 (SplitResidualVectorQuantizer.encode) / :1330-1333 (MimiModel.encode)."""
        K, nq, nsem = self.num_codebooks, self.config.num_quantizers, self.config.num_semantic_quantizers
        if K > nq:
            raise ValueError(
                f"The number of quantizers (i.e codebooks) asked should be lower than the total number of quantizers {nq}, but is currently {K}."
            )
        if K < nsem:
            raise ValueError(
                f"The number of quantizers (i.e codebooks) asked should be higher than the number of semantic quantizers {nsem}, but is currently {K}."
            )

    # override
    @torch.no_grad()
    def embs(self):
        nat = self._any_native()
        K = self.num_codebooks
        width = self.config.codebook_dim if self.latent else self.config.hidden_size
        out = torch.empty(K, self.vocab_size, width, device=nat.device)
        with torch.cuda.device(nat.device):
            fn = nat.lib.ac_embs if self.latent else nat.lib.ac_embs_projected
            _native.check(fn(nat.h, K, _ptr(out), _stream()), nat.h, "ac_embs")
        return out  # [K, C, D] (latent) or [K, C, hidden]

    # override
    def _sig_to_toks(self, sig, length):
        # sig: [B, T].  The padding mask the reference builds (mimi.py:95-104) is not applied to the
        # samples by the model ([HF] mimi :1245-1247): `length` does not change the result.
        self._check_num_codebooks()
        B, T = sig.shape
        K = self.num_codebooks
        N = self.config.num_frames(T)
        if B == 0:   # an empty shard (sharding.shard_bounds): nothing to run, the library is not called
            return torch.empty(0, N, K, dtype=torch.int64, device=sig.device)
        nat = self._native_for(sig)
        sig = sig.to(torch.float32).contiguous()
        toks = torch.empty(B, N, K, dtype=torch.int64, device=sig.device)
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_encode_workspace_bytes(nat.h, B, T))
            _native.check(
                nat.lib.ac_encode(nat.h, _ptr(sig), None, B, T, K, _ptr(toks), _ptr(ws), ws.numel(), _stream()),
                nat.h, "ac_encode",
            )
        return toks  # [B, N, K]

    # override
    def _sig_to_feats(self, sig, length):
        # sig: [B, T] -> [B, N, hidden]: encoder -> encoder_transformer -> downsample (mimi.py:112-121)
        B, T = sig.shape
        N = self.config.num_frames(T)
        if B == 0:
            return torch.empty(0, N, self.config.hidden_size, dtype=torch.float32, device=sig.device)
        nat = self._native_for(sig)
        sig = sig.to(torch.float32).contiguous()
        feats = torch.empty(B, N, self.config.hidden_size, dtype=torch.float32, device=sig.device)
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_encode_workspace_bytes(nat.h, B, T))
            _native.check(
                nat.lib.ac_encode_feats(nat.h, _ptr(sig), None, B, T, _ptr(feats), _ptr(ws), ws.numel(), _stream()),
                nat.h, "ac_encode_feats",
            )
        return feats

    # override
    def _sig_to_qfeats(self, sig, length):
        return self._toks_to_qfeats(self._sig_to_toks(sig, length), length)

    # override
    def _toks_to_sig(self, toks, length):
        # toks: [B, N, K] -> [B, N*hop] (not trimmed: mimi.py:146-148 passes no padding mask)
        B, N, K = toks.shape
        if B == 0:
            return torch.empty(0, N * self.config.hop_length, dtype=torch.float32, device=toks.device)
        nat = self._native_for(toks)
        toks = toks.to(torch.int64).contiguous()
        sig = torch.empty(B, N * self.config.hop_length, dtype=torch.float32, device=toks.device)
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_decode_workspace_bytes(nat.h, B, N))
            _native.check(
                nat.lib.ac_decode(nat.h, _ptr(toks), B, N, K, _ptr(sig), _ptr(ws), ws.numel(), _stream()),
                nat.h, "ac_decode",
            )
        return sig

    # override
    def _toks_to_qfeats(self, toks, length):
        # toks: [B, N, K] -> [B, N, hidden]
        B, N, K = toks.shape
        if B == 0:
            return torch.empty(0, N, self.config.hidden_size, dtype=torch.float32, device=toks.device)
        nat = self._native_for(toks)
        toks = toks.to(torch.int64).contiguous()
        out = torch.empty(B, N, self.config.hidden_size, dtype=torch.float32, device=toks.device)
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_quantizer_workspace_bytes(nat.h, B, N))
            _native.check(
                nat.lib.ac_dequantize_ws(nat.h, _ptr(toks), B, N, K, _ptr(out), _ptr(ws), ws.numel(), _stream()),
                nat.h, "ac_dequantize_ws",
            )
        return out

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
