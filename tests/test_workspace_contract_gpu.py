"""The boundary's memory contract (include/audiocodecs_amd.h, SURVEY.md section 8(b): "caller owns all tensors ... must not
sync internally"): with a workspace the caller allocated once, the raw C-ABI entry points neither allocate nor free device
memory while the batch size grows and shrinks from call to call (round 2 grew a handle-owned amax pool with
hipStreamSynchronize + hipFree + hipMalloc inside ac_encode / ac_decode), the reported workspace size is sufficient and a
byte less is refused, and results do not depend on what the workspace held before."""
import ctypes as C

import numpy as np
import pytest
import torch

from golden_cases import noise

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def enc(checkpoints):
    from audiocodecs_amd import Encodec

    cfg, sd = checkpoints("full", 0)
    codec = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    codec.sig_to_toks(noise(3, 1, 640).cuda())
    return codec, next(iter(codec._natives.values()))


def test_calls_with_a_growing_batch_are_graph_capturable(enc, checkpoints, monkeypatch):
    """hipStreamBeginCapture refuses hipMalloc / hipFree / hipStreamSynchronize: a sequence of raw ac_encode / ac_decode calls
    whose batch grows past everything the handle has seen (round 2 reallocated a handle-owned pool exactly there) is captured
    into ONE hipGraph from a workspace allocated beforehand, and its replays -- also after unrelated eager work in between --
    reproduce the eager results bit for bit.  Under capture the LSTM runs as per-step kernels (lstm_fwd: a replayed cooperative
    launch is not guaranteed its placement), so the expectation comes from a handle running the same per-step kernels."""
    from audiocodecs_amd import Encodec, _native

    codec, nat = enc
    L = nat.lib
    T, K, Bmax = 4800, 8, 40
    N = codec.config.num_frames(T)
    ws_bytes = max(L.ac_encode_workspace_bytes(nat.h, Bmax, T), L.ac_decode_workspace_bytes(nat.h, Bmax, N))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    sig = noise(11, Bmax, T).cuda()
    sizes = (2, 7, 40, 3)
    cfg, sd = checkpoints("full", 0)
    monkeypatch.setenv("AC_LSTM", "step")
    stepper = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    want = {}
    for B in sizes:
        t = stepper.sig_to_toks(sig[:B])
        want[B] = (t, stepper.toks_to_sig(t))
    toks = {B: torch.empty(B, N, K, dtype=torch.int64, device="cuda") for B in sizes}
    rec = {B: torch.empty(B, N * 320, device="cuda") for B in sizes}
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):                 # the handle has seen batches of 1 and 2 so far
            for B in sizes:
                _native.check(L.ac_encode(nat.h, _ptr(sig), None, B, T, K, _ptr(toks[B]), _ptr(ws), ws_bytes, _stream()), nat.h, "ac_encode")
                _native.check(L.ac_decode(nat.h, _ptr(toks[B]), B, N, K, _ptr(rec[B]), _ptr(ws), ws_bytes, _stream()), nat.h, "ac_decode")
    for rep in range(3):
        for B in sizes:
            toks[B].zero_(); rec[B].zero_()
        g.replay()
        torch.cuda.synchronize()
        for B in sizes:
            assert torch.equal(toks[B], want[B][0]), (rep, B)
            assert torch.equal(rec[B], want[B][1]), (rep, B)
        codec.toks_to_sig(codec.sig_to_toks(sig[: 5 + rep]))     # eager work on the same handle between replays (persistent LSTM, other sizes)
    assert nat.lib.ac_lstm_status(nat.h) == 1                    # ... which still runs the persistent kernel, and no launch ever failed


def test_reported_size_is_enough_and_one_byte_less_is_refused(enc):
    codec, nat = enc
    L = nat.lib
    B, T, K = 3, 2400, 8
    N = codec.config.num_frames(T)
    sig = noise(12, B, T).cuda()
    toks = torch.empty(B, N, K, dtype=torch.int64, device="cuda")
    need = L.ac_encode_workspace_bytes(nat.h, B, T)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    assert L.ac_encode(nat.h, _ptr(sig), None, B, T, K, _ptr(toks), _ptr(ws), need - 1, _stream()) == -3      # AC_ENOMEM
    assert b"workspace too small" in L.ac_last_error(nat.h)
    assert L.ac_encode(nat.h, _ptr(sig), None, B, T, K, _ptr(toks), _ptr(ws), need, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(toks, codec.sig_to_toks(sig))


def test_results_do_not_depend_on_the_workspace_contents(enc):
    """The pool inside the workspace is cleared by the call itself (one memset on the caller's stream): a workspace full of
    0xFF bytes -- NaN amax words -- gives the same tokens and waveform as a zeroed one."""
    codec, nat = enc
    L = nat.lib
    B, T, K = 4, 3200, 8
    N = codec.config.num_frames(T)
    sig = noise(13, B, T).cuda()
    ws_bytes = max(L.ac_encode_workspace_bytes(nat.h, B, T), L.ac_decode_workspace_bytes(nat.h, B, N))
    outs = []
    for fill in (0x00, 0xFF, 0x7F):
        ws = torch.full((ws_bytes,), fill, dtype=torch.uint8, device="cuda")
        toks = torch.empty(B, N, K, dtype=torch.int64, device="cuda")
        rec = torch.empty(B, N * 320, device="cuda")
        assert L.ac_encode(nat.h, _ptr(sig), None, B, T, K, _ptr(toks), _ptr(ws), ws_bytes, _stream()) == 0
        assert L.ac_decode(nat.h, _ptr(toks), B, N, K, _ptr(rec), _ptr(ws), ws_bytes, _stream()) == 0
        torch.cuda.synchronize()
        outs.append((toks, rec))
    for toks, rec in outs[1:]:
        assert torch.equal(toks, outs[0][0]) and torch.equal(rec, outs[0][1])


@pytest.mark.parametrize("name", ["mimi", "dac", "wavtokenizer"])
def test_other_codecs_are_graph_capturable_at_a_new_batch_size(name, mimi_checkpoints, dac_checkpoints, wavtok_checkpoints):
    """Same contract for the other three handle kinds (row ring for the linear layers of Mimi / WavTokenizer, amax slots per
    chunk for DAC): encode + decode at a batch size the handle has never seen is captured into a hipGraph (the wrapper's
    workspace tensor comes from torch's graph-safe allocator) and the replay equals the eager result."""
    from audiocodecs_amd import DAC, Mimi, WavTokenizer

    if name == "mimi":
        cfg, sd = mimi_checkpoints("tiny", 0)
        codec = Mimi(cfg.sampling_rate, num_codebooks=4, state_dict=sd, config=cfg).eval()
        T = 9600
    elif name == "dac":
        cfg, sd = dac_checkpoints("tiny", 0)
        codec = DAC(cfg.sampling_rate, cfg.sampling_rate, num_codebooks=cfg.n_codebooks, state_dict=sd, config=cfg).eval()
        T = 8192
    else:
        cfg, sd = wavtok_checkpoints("tiny", 0)
        codec = WavTokenizer(cfg.sampling_rate, state_dict=sd, arch=cfg).eval()
        T = 9600
    big = noise(21, 9, T).cuda()
    codec.toks_to_sig(codec.sig_to_toks(big[:2]))         # creates the handle; it has seen a batch of 2 only
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            toks = codec.sig_to_toks(big)                 # 9 clips: past anything seen
            rec = codec.toks_to_sig(toks)
    g.replay()
    torch.cuda.synchronize()
    et = codec.sig_to_toks(big)
    assert torch.equal(toks, et) and torch.equal(rec, codec.toks_to_sig(et))
