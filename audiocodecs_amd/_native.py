"""ctypes binding of libaudiocodecs_amd.so (include/audiocodecs_amd.h).

The shared library is the product; there is NO fallback: if it is missing, or no gfx950 device is
visible, every compute entry point raises.  `import torch` must precede the load so that the HIP
runtime torch already mapped (libamdhip64.so.7) is the one the library binds to.
"""

from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads libamdhip64 first)

__all__ = ["lib", "lib_path", "AcConfig", "AcMimiConfig", "AcDacConfig", "AcWavtokConfig", "AcKernelStat", "NativeError", "check", "EXPORTS", "track", "set_precision", "check_precision", "PRECISIONS"]

AC_MAX_RATIOS = 8
# AUDIOCODECS_AMD_LIB: developer override (timing variants built by hand); the product is the in-tree library
lib_path = os.environ.get("AUDIOCODECS_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libaudiocodecs_amd.so")


class NativeError(RuntimeError):
    pass


class AcConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("sampling_rate", C.c_int32),
        ("num_filters", C.c_int32),
        ("hidden_size", C.c_int32),
        ("num_ratios", C.c_int32),
        ("upsampling_ratios", C.c_int32 * AC_MAX_RATIOS),
        ("kernel_size", C.c_int32),
        ("last_kernel_size", C.c_int32),
        ("residual_kernel_size", C.c_int32),
        ("compress", C.c_int32),
        ("num_lstm_layers", C.c_int32),
        ("codebook_size", C.c_int32),
        ("num_quantizers", C.c_int32),
        ("device", C.c_int32),
    ]


class AcMimiConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("sampling_rate", C.c_int32),
        ("num_filters", C.c_int32),
        ("hidden_size", C.c_int32),
        ("num_ratios", C.c_int32),
        ("upsampling_ratios", C.c_int32 * AC_MAX_RATIOS),
        ("kernel_size", C.c_int32),
        ("last_kernel_size", C.c_int32),
        ("residual_kernel_size", C.c_int32),
        ("compress", C.c_int32),
        ("codebook_size", C.c_int32),
        ("codebook_dim", C.c_int32),
        ("num_quantizers", C.c_int32),
        ("num_semantic_quantizers", C.c_int32),
        ("num_hidden_layers", C.c_int32),
        ("num_attention_heads", C.c_int32),
        ("head_dim", C.c_int32),
        ("intermediate_size", C.c_int32),
        ("sliding_window", C.c_int32),
        ("resample_stride", C.c_int32),
        ("device", C.c_int32),
        ("rope_theta", C.c_float),
        ("norm_eps", C.c_float),
    ]


AC_MAX_DILATIONS = 4


class AcDacConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("sampling_rate", C.c_int32),
        ("encoder_hidden_size", C.c_int32),
        ("decoder_hidden_size", C.c_int32),
        ("num_ratios", C.c_int32),
        ("downsampling_ratios", C.c_int32 * AC_MAX_RATIOS),
        ("upsampling_ratios", C.c_int32 * AC_MAX_RATIOS),
        ("n_codebooks", C.c_int32),
        ("codebook_size", C.c_int32),
        ("codebook_dim", C.c_int32),
        ("num_dilations", C.c_int32),
        ("dilations", C.c_int32 * AC_MAX_DILATIONS),
        ("device", C.c_int32),
    ]


class AcWavtokConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("sampling_rate", C.c_int32),
        ("num_filters", C.c_int32),
        ("dimension", C.c_int32),
        ("num_ratios", C.c_int32),
        ("ratios", C.c_int32 * AC_MAX_RATIOS),
        ("kernel_size", C.c_int32),
        ("last_kernel_size", C.c_int32),
        ("residual_kernel_size", C.c_int32),
        ("compress", C.c_int32),
        ("num_lstm_layers", C.c_int32),
        ("codebook_size", C.c_int32),
        ("backbone_dim", C.c_int32),
        ("intermediate_dim", C.c_int32),
        ("num_layers", C.c_int32),
        ("adanorm_num_embeddings", C.c_int32),
        ("num_groups", C.c_int32),
        ("n_fft", C.c_int32),
        ("bandwidth_id", C.c_int32),
        ("device", C.c_int32),
    ]


class AcKernelStat(C.Structure):
    _fields_ = [
        ("name", C.c_char * 96),
        ("launches", C.c_int32),
        ("total_ms", C.c_float),
        ("flops", C.c_double),
        ("bytes", C.c_double),
    ]


_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
# name -> (restype, argtypes): every symbol include/audiocodecs_amd.h declares
EXPORTS = {
    "ac_version": (_i, []),
    "ac_create": (_i, [C.POINTER(AcConfig), C.POINTER(_vp)]),
    "ac_mimi_create": (_i, [C.POINTER(AcMimiConfig), C.POINTER(_vp)]),
    "ac_dac_create": (_i, [C.POINTER(AcDacConfig), C.POINTER(_vp)]),
    "ac_wavtok_create": (_i, [C.POINTER(AcWavtokConfig), C.POINTER(_vp)]),
    "ac_decode_feats": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ac_load_weights": (_i, [_vp, C.c_char_p, _vp, _sz]),
    "ac_set_precision": (_i, [_vp, _i]),
    "ac_finalize": (_i, [_vp]),
    "ac_num_frames": (_i, [_vp, _i]),
    "ac_num_samples": (C.c_longlong, [_vp, _i]),
    "ac_hop_length": (_i, [_vp]),
    "ac_hidden_size": (_i, [_vp]),
    "ac_codebook_dim": (_i, [_vp]),
    "ac_encode_workspace_bytes": (_sz, [_vp, _i, _i]),
    "ac_decode_workspace_bytes": (_sz, [_vp, _i, _i]),
    "ac_encode": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ac_encode_feats": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ac_decode": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ac_quantize": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ac_dequantize": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ac_encode_quantized": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ac_encode_feats_latent": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ac_quantizer_workspace_bytes": (_sz, [_vp, _i, _i]),
    "ac_quantize_ws": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ac_dequantize_ws": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ac_embs": (_i, [_vp, _i, _vp, _vp]),
    "ac_embs_projected": (_i, [_vp, _i, _vp, _vp]),
    "ac_resample": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "ac_profile_begin": (_i, [_vp]),
    "ac_profile_end": (_i, [_vp, C.POINTER(AcKernelStat), _i]),
    "ac_debug_clock": (_i, [_vp, _i, C.POINTER(C.c_double)]),
    "ac_debug_bounds": (_i, [_vp, C.POINTER(C.c_float), _i]),
    "ac_debug_trace": (_i, [_vp, C.POINTER(C.c_ulonglong), _i]),
    "ac_debug_split_row": (_i, [_vp, _i, _vp, _vp]),
    "ac_debug_capture": (_i, [_vp, _vp, _sz]),
    "ac_debug_set": (_i, [_vp, C.c_char_p, _i]),
    "ac_debug_captured": (_sz, [_vp]),
    "ac_lstm_status": (_i, [_vp]),
    "ac_poll_status": (_i, [_vp, _vp]),
    "ac_last_error": (C.c_char_p, [_vp]),
    "ac_destroy": (None, [_vp]),
}

PRECISIONS = {"fp32": 0, "fp32_exact": 1}   # AC_PRECISION_*


def check_precision(precision):
    if precision is not None and precision not in PRECISIONS:
        raise ValueError(f"`precision` ({precision}) must be one of {list(PRECISIONS)}")
    return precision


def set_precision(L, h, precision) -> None:
    """precision: None (library default / AC_GEMM), "fp32" (split16: fp32 fidelity on the fp16 matrix pipe), "fp32_exact" (IEEE fp32 products)."""
    if precision is None:
        return
    if precision not in PRECISIONS:
        raise ValueError(f"`precision` ({precision}) must be one of {list(PRECISIONS)}")
    check(L.ac_set_precision(h, PRECISIONS[precision]), h, "ac_set_precision")


def debug_set(codec, key: str, value: int) -> None:
    """Developer / test switch on every live handle of a wrapper (include/audiocodecs_amd.h ac_debug_set).  Handles are created
    lazily per device: call the wrapper (or `codec._native_for(tensor)`) first.  Switches start from the environment variables of
    the same meaning, which the library reads once per handle at ac_finalize."""
    if not codec._natives:
        raise NativeError("debug_set: the wrapper has no handle yet (call it, or _native_for(tensor), first)")
    for nat in codec._natives.values():
        check(nat.lib.ac_debug_set(nat.h, key.encode(), int(value)), nat.h, "ac_debug_set")


_lib = None
_live = None   # weak set of objects holding an ac_handle (attribute `h`): destroyed at interpreter exit, while the HIP
               # runtime is still up (a handle owns device memory, pinned host memory and events)


def track(obj) -> None:
    """Register an object with attributes `lib` and `h` (an ac_handle) for destruction at exit."""
    global _live
    if _live is None:
        import atexit
        import weakref

        _live = weakref.WeakSet()

        def _destroy_all():
            for o in list(_live):
                try:
                    if getattr(o, "h", None):
                        o.lib.ac_destroy(o.h)
                        o.h = None
                except Exception:
                    pass

        atexit.register(_destroy_all)
    _live.add(obj)



def lib():
    """Load (once) and return the library; raises NativeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path):
            raise NativeError(
                f"{lib_path} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or audiocodecs_amd/csrc/build.sh)"
            )
        L = C.CDLL(lib_path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, handle=None, what: str = "") -> None:
    if rc < 0:
        msg = lib().ac_last_error(handle).decode() if handle else ""
        raise NativeError(f"{what} failed with code {rc}: {msg}")
