"""EnCodec on MI355X -- host-side mirror of the reference wrapper `audiocodecs.Encodec`
(/root/reference/audiocodecs/encodec.py:30-149): same constructor arguments, attributes
(`num_codebooks`, `vocab_size`, `bandwidth`), method names, tensor layouts and error behaviour.
The third-party `transformers.EncodecModel` the reference calls (encodec.py:51,90,116,125,139,147)
is replaced by the hand-written gfx950 kernels behind the C ABI in include/audiocodecs_amd.h.
PyTorch is used here only for device memory, streams and one-time weight-norm folding.
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _native, checkpoint
from .codec import Codec
from .config import ENCODEC_24KHZ, EncodecConfig

__all__ = ["Encodec"]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Native:
    """One ac_handle: weights on one GPU + a grow-only workspace tensor."""

    def __init__(self, cfg: EncodecConfig, folded: Dict[str, torch.Tensor], device: torch.device, precision=None):
        self.lib = _native.lib()
        c = _native.AcConfig()
        c.struct_size = C.sizeof(_native.AcConfig)
        c.sampling_rate = cfg.sampling_rate
        c.num_filters = cfg.num_filters
        c.hidden_size = cfg.hidden_size
        c.num_ratios = len(cfg.upsampling_ratios)
        for i, r in enumerate(cfg.upsampling_ratios):
            c.upsampling_ratios[i] = r
        c.kernel_size = cfg.kernel_size
        c.last_kernel_size = cfg.last_kernel_size
        c.residual_kernel_size = cfg.residual_kernel_size
        c.compress = cfg.compress
        c.num_lstm_layers = cfg.num_lstm_layers
        c.codebook_size = cfg.codebook_size
        c.num_quantizers = cfg.num_quantizers
        c.device = device.index if device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", c.device)
        self.h = C.c_void_p()
        rc = self.lib.ac_create(C.byref(c), C.byref(self.h))
        if rc < 0:
            raise _native.NativeError(f"ac_create failed with code {rc} (is a gfx950 GPU visible?)")
        _native.set_precision(self.lib, self.h, precision)
        for name, t in folded.items():
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
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
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


class Encodec(Codec):
    _accepts_none_length = True
    _graph_capable = False        # codec.py: the persistent LSTM launch is not replayable from a hipGraph

    def __init__(
        self,
        sample_rate,
        orig_sample_rate=24000,
        mode="reconstruct",
        num_codebooks=8,
        use_vocos=False,
        *,
        state_dict: Optional[Dict[str, torch.Tensor]] = None,
        config: EncodecConfig = ENCODEC_24KHZ,
        precision: Optional[str] = None,
        strict: bool = False,
        graph: bool = False,
    ):
        """`state_dict`: an HF-format EncodecModel state dict (keys of SURVEY.md Appendix A.3, e.g.
        `safetensors.torch.load_file(model.safetensors)` of facebook/encodec_24khz, or
        `checkpoint.synthetic_state_dict(cfg, seed)`).  When omitted the pretrained checkpoint is
        fetched through huggingface_hub like the reference does (needs network or a warm cache).
        `precision`: None / "fp32" = fp32 fidelity on the fp16 matrix pipe (split16: the parity arithmetic, default);
        "fp32_exact" = exact fp32 products (include/audiocodecs_amd.h ac_set_precision)."""
        super().__init__(sample_rate, orig_sample_rate, mode)
        self.strict = bool(strict)   # codec.py: poll the handle after every call
        self.graph = bool(graph)     # codec.py: replay one hipGraph per (call, shape)
        self.precision = _native.check_precision(precision)
        if use_vocos:
            raise NotImplementedError("the Vocos decoder variant (encodec.py:53-66) is outside the MI355X path")
        if config.sampling_rate != orig_sample_rate:
            raise ValueError(f"config.sampling_rate ({config.sampling_rate}) != orig_sample_rate ({orig_sample_rate})")
        self.num_codebooks = num_codebooks
        self.use_vocos = use_vocos
        self.vocab_size = config.codebook_size
        self.config = config
        self.bandwidth = (num_codebooks * 75) / 100  # encodec.py:50
        if state_dict is None:
            state_dict = self._fetch_pretrained(int(orig_sample_rate / 1000))
        self._folded = checkpoint.fold_weight_norm(state_dict)
        # encodec.py:67-71: the half of the model the mode never runs is dropped (here: never packed or uploaded)
        if mode == "encode":
            self._folded = {k: v for k, v in self._folded.items() if not k.startswith("decoder.")}
        elif mode == "decode":
            self._folded = {k: v for k, v in self._folded.items() if not k.startswith("encoder.")}
        self._natives: Dict[int, _Native] = {}

    @staticmethod
    def _fetch_pretrained(tag: int):
        try:
            from huggingface_hub import hf_hub_download
            from safetensors.torch import load_file
        except ImportError:
            raise ImportError("`pip install huggingface_hub safetensors` to fetch pretrained EnCodec weights")
        return load_file(hf_hub_download(f"facebook/encodec_{tag}khz", "model.safetensors"))

    # ------------------------------------------------------------------------------------------
    def _native_for(self, t: torch.Tensor) -> _Native:
        if not t.is_cuda:
            raise _native.NativeError(
                "audiocodecs_amd runs on MI355X only: move the input to a cuda device "
                "(there is deliberately no CPU fallback)"
            )
        idx = t.device.index
        if idx not in self._natives:
            self._natives[idx] = _Native(self.config, self._folded, t.device, self.precision)
        return self._natives[idx]

    def _num_quantizers(self) -> int:
        """[HF] modeling_encodec.py:564-567 rejects bandwidths outside config.target_bandwidths,
        then :416-422 maps the bandwidth to a stage count."""
        if self.bandwidth not in self.config.target_bandwidths:
            raise ValueError(
                f"This model doesn't support the bandwidth {self.bandwidth}. "
                f"Select one of {list(self.config.target_bandwidths)}."
            )
        return self.config.num_quantizers_for_bandwidth(self.bandwidth)

    def _check_length(self, sig, length):
        """encodec.py:84-89 builds a [B, max_len] mask with max_len = int(max(T*length)); the model
        then multiplies it with the [B,1,T] input, which only works when max_len == T."""
        if length is None:
            return None
        length = length.to(device=sig.device, dtype=torch.float32).contiguous()
        max_len = int((sig.shape[-1] * length).max().long().item())
        if max_len != sig.shape[-1]:
            raise RuntimeError(
                f"The size of the padding mask ({max_len}) must match the signal length ({sig.shape[-1]}): "
                "relative lengths must have a maximum of 1.0"
            )
        return length

    # override
    @torch.no_grad()
    def embs(self):
        dev = next(iter(self._natives.values())).device if self._natives else torch.device("cuda", torch.cuda.current_device())
        nat = self._native_for(torch.empty(0, device=dev))
        out = torch.empty(self.num_codebooks, self.vocab_size, self.config.hidden_size, device=nat.device)
        with torch.cuda.device(nat.device):
            _native.check(nat.lib.ac_embs(nat.h, self.num_codebooks, _ptr(out), _stream()), nat.h, "ac_embs")
        return out  # [K, C, H]

    # override
    def _sig_to_toks(self, sig, length):
        # sig: [B, T]
        K = self._num_quantizers()
        B, T = sig.shape
        N = self.config.num_frames(T)
        if B == 0:   # an empty shard (sharding.shard_bounds): nothing to run, the library is not called
            return torch.empty(0, N, K, dtype=torch.int64, device=sig.device)
        nat = self._native_for(sig)
        sig = sig.to(torch.float32).contiguous()
        length = self._check_length(sig, length)
        toks = torch.empty(B, N, K, dtype=torch.int64, device=sig.device)
        with torch.cuda.device(nat.device):
            nbytes = nat.lib.ac_encode_workspace_bytes(nat.h, B, T)
            ws = nat.workspace(nbytes)
            _native.check(
                nat.lib.ac_encode(nat.h, _ptr(sig), _ptr(length), B, T, K, _ptr(toks), _ptr(ws), ws.numel(), _stream()),
                nat.h, "ac_encode",
            )
        return toks  # [B, N, K]

    # override
    def _sig_to_feats(self, sig, length):
        # sig: [B, T] -> [B, N, H].  The reference masks here only when config.normalize
        # (encodec.py:107-112): never for the 24 kHz model, so `length` is ignored.
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
        toks = self._sig_to_toks(sig, length)
        return self._toks_to_qfeats(toks, length)

    # override
    def _toks_to_sig(self, toks, length):
        # toks: [B, N, K] -> [B, N*hop]
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
        # toks: [B, N, K] -> [B, N, H]
        B, N, K = toks.shape
        if B == 0:
            return torch.empty(0, N, self.config.hidden_size, dtype=torch.float32, device=toks.device)
        nat = self._native_for(toks)
        toks = toks.to(torch.int64).contiguous()
        out = torch.empty(B, N, self.config.hidden_size, dtype=torch.float32, device=toks.device)
        with torch.cuda.device(nat.device):
            _native.check(nat.lib.ac_dequantize(nat.h, _ptr(toks), B, N, K, _ptr(out), _stream()), nat.h, "ac_dequantize")
        return out

    # ---- measurement hook used by bench.py ------------------------------------------------------
    def profile_kernels(self, fn):
        """Run fn() with per-kernel HIP-event timing armed; returns [(name, launches, ms, flops, bytes)]."""
        nat = self._native_for(torch.empty(0, device=torch.device("cuda", torch.cuda.current_device())))
        _native.check(nat.lib.ac_profile_begin(nat.h), nat.h, "ac_profile_begin")
        try:
            fn()
        finally:
            buf = (_native.AcKernelStat * 256)()
            n = nat.lib.ac_profile_end(nat.h, buf, 256)
        _native.check(n, nat.h, "ac_profile_end")
        return [(buf[i].name.decode(), buf[i].launches, buf[i].total_ms, buf[i].flops, buf[i].bytes) for i in range(n)]
