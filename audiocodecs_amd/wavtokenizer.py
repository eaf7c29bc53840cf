"""WavTokenizer on MI355X -- host-side mirror of the reference wrapper `audiocodecs.WavTokenizer`
(/root/reference/audiocodecs/wavtokenizer.py:31-135): same constructor arguments (`sample_rate`, `mode`, `source`,
`config`, `checkpoint`), attributes (`num_codebooks` = 1, `vocab_size` = 4096), method names, tensor layouts.
The third-party package `wavtokenizer` the reference calls (wavtokenizer.py:58,76-78,87,94,101,108,115-118,130-133) is
replaced by the gfx950 kernels behind the C ABI (include/audiocodecs_amd.h, ac_wavtok_create).

PARITY UNPINNED: that package is not on disk here, so this path is pinned to oracle/wavtokenizer_oracle.py -- a
restatement of its published modules -- not to outputs of the reference itself.
PyTorch is used for device memory and streams only.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _native
from .codec import Codec
from .config import WAVTOK_40, WAVTOK_75, WavTokenizerConfig
from .encodec import _ptr, _stream

__all__ = ["WavTokenizer"]


class _NativeWavTok:
    """One WavTokenizer ac_handle: weights on one GPU + a grow-only workspace tensor."""

    def __init__(self, cfg: WavTokenizerConfig, sd: Dict[str, torch.Tensor], device: torch.device, precision=None):
        self.lib = _native.lib()
        c = _native.AcWavtokConfig()
        c.struct_size = C.sizeof(_native.AcWavtokConfig)
        for f in ("sampling_rate", "num_filters", "dimension", "kernel_size", "last_kernel_size", "residual_kernel_size", "compress",
                  "num_lstm_layers", "codebook_size", "backbone_dim", "intermediate_dim", "num_layers", "adanorm_num_embeddings",
                  "num_groups", "n_fft", "bandwidth_id"):
            setattr(c, f, getattr(cfg, f))
        c.num_ratios = len(cfg.ratios)
        for i, r in enumerate(cfg.ratios):
            c.ratios[i] = r
        c.device = device.index if device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", c.device)
        self.h = C.c_void_p()
        rc = self.lib.ac_wavtok_create(C.byref(c), C.byref(self.h))
        if rc < 0:
            raise _native.NativeError(f"ac_wavtok_create failed with code {rc} (unsupported configuration, or no gfx950 GPU visible)")
        _native.set_precision(self.lib, self.h, precision)
        for name, t in sd.items():
            if not torch.is_tensor(t) or not t.is_floating_point():
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

            if sys.is_finalizing():
                return
            if getattr(self, "h", None):
                self.lib.ac_destroy(self.h)
                self.h = None
        except Exception:
            pass


class WavTokenizer(Codec):
    _accepts_none_length = True
    _graph_capable = False        # codec.py: the persistent LSTM launch is not replayable from a hipGraph
    SOURCES = [
        "novateur/WavTokenizer-large-unify-40token",
        "novateur/WavTokenizer-large-speech-75token",
    ]
    CONFIGS = [
        "wavtokenizer_smalldata_frame40_3s_nq1_code4096_dim512_kmeans200_attn.yaml",
        "wavtokenizer_smalldata_frame75_3s_nq1_code4096_dim512_kmeans200_attn.yaml",
    ]
    CHECKPOINTS = [
        "wavtokenizer_large_unify_600_24k.ckpt",
        "wavtokenizer_large_speech_320_v2.ckpt",
    ]

    def __init__(
        self,
        sample_rate,
        mode="reconstruct",
        source="novateur/WavTokenizer-large-unify-40token",
        config="wavtokenizer_smalldata_frame40_3s_nq1_code4096_dim512_kmeans200_attn.yaml",
        checkpoint="wavtokenizer_large_unify_600_24k.ckpt",
        *,
        state_dict: Optional[Dict[str, torch.Tensor]] = None,
        arch: Optional[WavTokenizerConfig] = None,
        precision: Optional[str] = None,
        strict: bool = False,
        graph: bool = False,
    ):
        """`state_dict`: the `state_dict` of the upstream Lightning checkpoint (keys feature_extractor.* / backbone.* /
        head.*), or `checkpoint.synthetic_wavtok_state_dict(arch, seed)`; when omitted it is fetched through
        huggingface_hub like the reference does (wavtokenizer.py:72-78; needs network or a warm cache).  `arch`: the
        architecture the YAML `config` describes (default: picked from the config's name -- frame40 / frame75)."""
        super().__init__(sample_rate, 24000, mode)  # wavtokenizer.py:68
        self.strict = bool(strict)   # codec.py: poll the handle after every call
        self.graph = bool(graph)     # codec.py: replay one hipGraph per (call, shape)
        if arch is None:
            arch = WAVTOK_75 if "frame75" in config else WAVTOK_40
        self.num_codebooks = 1
        self.vocab_size = arch.codebook_size  # 4096 (wavtokenizer.py:70)
        self.arch = arch
        self.precision = _native.check_precision(precision)   # see Encodec: None / "fp32" (parity arithmetic), "fp32_exact"
        if state_dict is None:
            state_dict = self._fetch_pretrained(source, checkpoint)
        # Upstream's from_pretrained0802 keeps the generator only; a Lightning checkpoint also carries the discriminators and the
        # loss modules (hundreds of MB that would be copied to the host twice per handle just to be dropped), and the encodec
        # DECODER inside the feature extractor is never used for inference
        keep = ("feature_extractor.encodec.encoder.", "feature_extractor.encodec.quantizer.", "backbone.", "head.")
        sd = {k: v for k, v in state_dict.items() if k.startswith(keep)}
        # wavtokenizer.py:80-84: the half the mode never runs is dropped (here: never packed or uploaded)
        if mode == "encode":
            sd = {k: v for k, v in sd.items() if not (k.startswith("backbone.") or k.startswith("head."))}
        elif mode == "decode":
            sd = {k: v for k, v in sd.items() if not k.startswith("feature_extractor.encodec.encoder.")}
        self._sd = sd
        self._natives: Dict[int, _NativeWavTok] = {}

    @staticmethod
    def _fetch_pretrained(source: str, checkpoint: str):
        try:
            from huggingface_hub import snapshot_download
        except ImportError:
            raise ImportError("`pip install huggingface_hub` to fetch pretrained WavTokenizer weights")
        path = os.path.join(snapshot_download(repo_id=source), checkpoint)
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        return ckpt.get("state_dict", ckpt)

    # ------------------------------------------------------------------------------------------
    def _native_for(self, t: torch.Tensor) -> _NativeWavTok:
        if not t.is_cuda:
            raise _native.NativeError(
                "audiocodecs_amd runs on MI355X only: move the input to a cuda device "
                "(there is deliberately no CPU fallback)"
            )
        idx = t.device.index
        if idx not in self._natives:
            self._natives[idx] = _NativeWavTok(self.arch, self._sd, t.device, self.precision)
        return self._natives[idx]

    def _any_native(self) -> _NativeWavTok:
        dev = next(iter(self._natives.values())).device if self._natives else torch.device("cuda", torch.cuda.current_device())
        return self._native_for(torch.empty(0, device=dev))

    # override
    @torch.no_grad()
    def embs(self):
        nat = self._any_native()
        out = torch.empty(1, self.vocab_size, self.arch.dimension, device=nat.device)
        with torch.cuda.device(nat.device):
            _native.check(nat.lib.ac_embs(nat.h, 1, _ptr(out), _stream()), nat.h, "ac_embs")
        return out  # [K=1, C, H]

    # override
    def _sig_to_toks(self, sig, length):
        # sig: [B, T] -> [B, N, 1]  (`length` is not used by the reference either, wavtokenizer.py:92-96)
        if sig.shape[0] == 0:   # an empty shard (sharding.shard_bounds): nothing to run, the library is not called
            return torch.empty(0, self.arch.num_frames(sig.shape[1]), 1, dtype=torch.int64, device=sig.device)
        nat = self._native_for(sig)
        sig = sig.to(torch.float32).contiguous()
        B, T = sig.shape
        N = self.arch.num_frames(T)
        toks = torch.empty(B, N, 1, dtype=torch.int64, device=sig.device)
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_encode_workspace_bytes(nat.h, B, T))
            _native.check(
                nat.lib.ac_encode(nat.h, _ptr(sig), None, B, T, 1, _ptr(toks), _ptr(ws), ws.numel(), _stream()),
                nat.h, "ac_encode",
            )
        return toks

    # override
    def _sig_to_feats(self, sig, length):
        # sig: [B, T] -> [B, N, dimension]
        if sig.shape[0] == 0:   # an empty shard (sharding.shard_bounds): nothing to run, the library is not called
            return torch.empty(0, self.arch.num_frames(sig.shape[1]), self.arch.dimension, dtype=torch.float32, device=sig.device)
        nat = self._native_for(sig)
        sig = sig.to(torch.float32).contiguous()
        B, T = sig.shape
        N = self.arch.num_frames(T)
        feats = torch.empty(B, N, self.arch.dimension, dtype=torch.float32, device=sig.device)
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_encode_workspace_bytes(nat.h, B, T))
            _native.check(
                nat.lib.ac_encode_feats(nat.h, _ptr(sig), None, B, T, _ptr(feats), _ptr(ws), ws.numel(), _stream()),
                nat.h, "ac_encode_feats",
            )
        return feats

    # override
    def _sig_to_qfeats(self, sig, length):
        # the quantised features `model.encode` returns are the selected code vectors (eval mode)
        return self._toks_to_qfeats(self._sig_to_toks(sig, length), length)

    # override
    def _toks_to_sig(self, toks, length):
        # toks: [B, N, 1] -> [B, N*hop]
        if toks.shape[0] == 0:   # an empty shard (sharding.shard_bounds): nothing to run, the library is not called
            return torch.empty(0, toks.shape[1] * self.arch.hop_length, dtype=torch.float32, device=toks.device)
        nat = self._native_for(toks)
        toks = toks.to(torch.int64).contiguous()
        B, N, K = toks.shape
        sig = torch.empty(B, N * self.arch.hop_length, dtype=torch.float32, device=toks.device)
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_decode_workspace_bytes(nat.h, B, N))
            _native.check(
                nat.lib.ac_decode(nat.h, _ptr(toks), B, N, K, _ptr(sig), _ptr(ws), ws.numel(), _stream()),
                nat.h, "ac_decode",
            )
        return sig

    # override
    def _toks_to_qfeats(self, toks, length):
        # toks: [B, N, 1] -> [B, N, dimension]
        if toks.shape[0] == 0:   # an empty shard (sharding.shard_bounds): nothing to run, the library is not called
            return torch.empty(0, toks.shape[1], self.arch.dimension, dtype=torch.float32, device=toks.device)
        nat = self._native_for(toks)
        toks = toks.to(torch.int64).contiguous()
        B, N, K = toks.shape
        out = torch.empty(B, N, self.arch.dimension, dtype=torch.float32, device=toks.device)
        with torch.cuda.device(nat.device):
            _native.check(nat.lib.ac_dequantize(nat.h, _ptr(toks), B, N, K, _ptr(out), _stream()), nat.h, "ac_dequantize")
        return out

    # override
    def _feats_to_sig(self, feats, length):
        # feats: [B, N, dimension] -> [B, N*hop]
        if feats.shape[0] == 0:   # an empty shard (sharding.shard_bounds): nothing to run, the library is not called
            return torch.empty(0, feats.shape[1] * self.arch.hop_length, dtype=torch.float32, device=feats.device)
        nat = self._native_for(feats)
        feats = feats.to(torch.float32).contiguous()
        B, N, H = feats.shape
        if H != self.arch.dimension:
            raise RuntimeError(f"expected features of width {self.arch.dimension}, got {H}")
        sig = torch.empty(B, N * self.arch.hop_length, dtype=torch.float32, device=feats.device)
        with torch.cuda.device(nat.device):
            ws = nat.workspace(nat.lib.ac_decode_workspace_bytes(nat.h, B, N))
            _native.check(
                nat.lib.ac_decode_feats(nat.h, _ptr(feats), B, N, _ptr(sig), _ptr(ws), ws.numel(), _stream()),
                nat.h, "ac_decode_feats",
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
