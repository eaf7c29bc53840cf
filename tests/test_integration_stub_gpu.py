"""The ctypes binding INTEGRATION.md section 2 shows a maintainer (struct ac_config by hand, ac_create / ac_load_weights with the
HF state-dict keys incl. the UNFOLDED weight-norm pairs / ac_finalize / ac_encode / ac_decode on raw pointers) -- executed as
written, against the shared library only, and compared with the packaged host class."""
import ctypes as C

import pytest
import torch

from golden_cases import noise

pytestmark = pytest.mark.gpu


class _Cfg(C.Structure):                          # struct ac_config, as in INTEGRATION.md
    _fields_ = [(n, C.c_int32) for n in ("struct_size", "sampling_rate", "num_filters", "hidden_size", "num_ratios")] \
             + [("upsampling_ratios", C.c_int32 * 8)] \
             + [(n, C.c_int32) for n in ("kernel_size", "last_kernel_size", "residual_kernel_size", "compress",
                                          "num_lstm_layers", "codebook_size", "num_quantizers", "device")]


def test_documented_ctypes_stub_runs_and_matches_the_host_class(checkpoints):
    from audiocodecs_amd import Encodec, _native

    cfg, sd = checkpoints("full", 0)
    lib = C.CDLL(_native.lib_path)
    lib.ac_encode_workspace_bytes.restype = C.c_size_t
    lib.ac_decode_workspace_bytes.restype = C.c_size_t
    lib.ac_last_error.restype = C.c_char_p
    c = _Cfg(C.sizeof(_Cfg), cfg.sampling_rate, cfg.num_filters, cfg.hidden_size, len(cfg.upsampling_ratios),
             (C.c_int32 * 8)(*cfg.upsampling_ratios), cfg.kernel_size, cfg.last_kernel_size, cfg.residual_kernel_size,
             cfg.compress, cfg.num_lstm_layers, cfg.codebook_size, cfg.num_quantizers, 0)
    h = C.c_void_p()
    assert lib.ac_create(C.byref(c), C.byref(h)) == 0
    for name, t in sd.items():                    # HF keys; (g, v) weight-norm pairs are folded inside ac_finalize
        if t.is_floating_point():
            t = t.detach().float().cpu().contiguous()
            assert lib.ac_load_weights(h, name.encode(), C.c_void_p(t.data_ptr()), C.c_size_t(t.numel() * 4)) == 0, name
    assert lib.ac_finalize(h) == 0, lib.ac_last_error(h)
    sig = noise(4545, 3, 20000).cuda()
    length = torch.tensor([1.0, 0.6, 0.9], device="cuda")
    B, T, K = 3, 20000, 8
    N = lib.ac_num_frames(h, T)
    toks = torch.empty(B, N, K, dtype=torch.int64, device="cuda")
    ws = torch.empty(lib.ac_encode_workspace_bytes(h, B, T), dtype=torch.uint8, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.ac_encode(h, C.c_void_p(sig.data_ptr()), C.c_void_p(length.data_ptr()), B, T, K, C.c_void_p(toks.data_ptr()),
                       C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), stream)
    assert rc == 0, lib.ac_last_error(h)
    rec = torch.empty(B, N * 320, dtype=torch.float32, device="cuda")
    ws2 = torch.empty(lib.ac_decode_workspace_bytes(h, B, N), dtype=torch.uint8, device="cuda")
    rc = lib.ac_decode(h, C.c_void_p(toks.data_ptr()), B, N, K, C.c_void_p(rec.data_ptr()), C.c_void_p(ws2.data_ptr()),
                       C.c_size_t(ws2.numel()), stream)
    assert rc == 0, lib.ac_last_error(h)
    torch.cuda.synchronize()
    codec = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    want = codec.sig_to_toks(sig, length)
    # the stub hands over UNFOLDED (g, v) pairs (folded in C++: v * (g / ||v||)); the class folds with torch._weight_norm:
    # the same weights up to one fp32 rounding -> the same tokens up to fp32 near-ties, the same waveform to 1e-5
    assert float((toks == want).float().mean()) > 0.999
    assert float((rec - codec.toks_to_sig(toks)).pow(2).mean().sqrt()) < 1e-5
    lib.ac_destroy(h)
