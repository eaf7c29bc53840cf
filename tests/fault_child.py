"""Child process of tests/test_fault_injection_gpu.py: the fault-injection cases of the persistent LSTM, run against the DEVELOPER library
(libaudiocodecs_amd_dev.so, AUDIOCODECS_AMD_LIB set by the parent) -- the product library carries no fault injection (csrc/split16.h
AC_DEV_MODE).  Usage: python tests/fault_child.py <case>"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from golden_cases import noise  # noqa: E402

_CKPT = {}


def checkpoints(cfg_name, seed):
    from audiocodecs_amd import checkpoint
    from audiocodecs_amd.config import ENCODEC_24KHZ, TINY

    key = (cfg_name, seed)
    if key not in _CKPT:
        cfg = {"full": ENCODEC_24KHZ, "tiny": TINY}[cfg_name]
        _CKPT[key] = (cfg, checkpoint.synthetic_state_dict(cfg, seed=seed))
    return _CKPT[key]


def test_failed_persistent_launch_is_reported_and_healed():
    """A persistent LSTM launch that fails (bounded wait expired / XCD placement broken; forced here by the kernel's test
    hook AC_LSTM_DBG=16) must never hand back unwritten memory: the tail kernel sets the launch's outputs to NaN and raises
    a sticky word; the NEXT call on the handle returns AC_EHIP once and the handle switches to the per-step kernels, after
    which it produces the same tokens as a healthy handle."""
    from audiocodecs_amd import Encodec
    from audiocodecs_amd._native import NativeError, debug_set

    cfg, sd = checkpoints("full", 0)
    good = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    codec = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    sig = noise(8282, 3, 16000).cuda()
    want = good.sig_to_toks(sig)
    codec._native_for(sig)
    debug_set(codec, "lstm_dbg", 16)         # fault injection: the next persistent launch reports a broken placement
    feats = codec.sig_to_feats(sig)          # the call itself cannot know: nothing synchronises
    torch.cuda.synchronize()
    debug_set(codec, "lstm_dbg", 0)
    assert bool(torch.isnan(feats).all())    # ... but its outputs are NaN, not garbage
    nat = next(iter(codec._natives.values()))
    assert nat.lib.ac_lstm_status(nat.h) < 0
    with pytest.raises(NativeError, match="persistent LSTM launch failed"):
        codec.sig_to_toks(sig)
    toks = codec.sig_to_toks(sig)            # healed: per-step kernels from now on
    assert nat.lib.ac_lstm_status(nat.h) == 0
    assert float((toks == want).float().mean()) > 0.999   # per-step vs persistent: same function up to fp32 rounding


def test_strict_mode_raises_in_the_call_that_failed():
    """strict=True (round-2 advisor finding on sticky errors): the wrapper polls the handle after its own call
    (ac_poll_status: synchronises the stream, reports and clears the sticky words), so the call whose persistent LSTM launch
    failed raises -- not an unrelated later one -- and the next call runs on the healed handle without an exception."""
    from audiocodecs_amd import Encodec
    from audiocodecs_amd._native import NativeError, debug_set

    cfg, sd = checkpoints("full", 0)
    good = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    codec = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg, strict=True).eval()
    sig = noise(8283, 2, 16000).cuda()
    want = good.sig_to_toks(sig)
    codec._native_for(sig)
    debug_set(codec, "lstm_dbg", 16)
    with pytest.raises(NativeError, match="persistent LSTM launch failed"):
        codec.sig_to_feats(sig)
    debug_set(codec, "lstm_dbg", 0)
    toks = codec.sig_to_toks(sig)            # no leftover error: the poll cleared it; per-step kernels from now on
    assert float((toks == want).float().mean()) > 0.999
    bad = toks.clone()
    bad[0, 3, 2] = 5000                      # outside [0, 1024)
    with pytest.raises(NativeError, match="token ids outside"):
        codec.toks_to_sig(bad)
    rec = codec.toks_to_sig(toks)            # unaffected
    assert bool(torch.isfinite(rec).all())


def test_one_poll_reports_and_clears_every_pending_failure():
    """Round-3 advisor finding: ac_poll_status reported one sticky class per call, so a bad-token count pending beside an LSTM
    failure surfaced in a later, unrelated call.  Both are raised in ONE message now and nothing is left behind."""
    from audiocodecs_amd import Encodec
    from audiocodecs_amd._native import NativeError, check, debug_set

    cfg, sd = checkpoints("full", 0)
    codec = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    sig = noise(8284, 4, 120000).cuda()
    toks = codec.sig_to_toks(sig)
    bad = toks.clone()
    bad[1, 7, 3] = 4096
    torch.cuda.synchronize()
    debug_set(codec, "lstm_dbg", 16)
    codec.toks_to_sig(bad)                   # ONE call raises two classes: out-of-range ids in the gather AND its persistent LSTM launch fails
    debug_set(codec, "lstm_dbg", 0)
    nat = next(iter(codec._natives.values()))
    stream = torch.cuda.current_stream().cuda_stream
    with pytest.raises(NativeError) as ei:
        check(nat.lib.ac_poll_status(nat.h, stream), nat.h, "ac_poll_status")
    msg = str(ei.value)
    assert "persistent LSTM launch failed" in msg and "token ids outside" in msg, msg
    check(nat.lib.ac_poll_status(nat.h, stream), nat.h, "ac_poll_status")      # nothing left behind
    assert bool(codec.toks_to_sig(codec.sig_to_toks(sig)).isfinite().all())      # healed handle, per-step LSTM kernels


if __name__ == "__main__":
    assert os.environ.get("AUDIOCODECS_AMD_LIB", "").endswith("_dev.so"), "run through tests/test_fault_injection_gpu.py (developer library)"
    globals()[sys.argv[1]]()
    print("FAULT_CHILD_OK", sys.argv[1])
