"""Fault injection into the persistent LSTM (a launch whose bounded waits expired / whose XCD placement broke) lives in the DEVELOPER
library only: the product library (libaudiocodecs_amd.so) carries neither the kernels' test hooks nor the switches that arm them
(csrc/split16.h AC_DEV_MODE).  The cases therefore run in a child process that loads libaudiocodecs_amd_dev.so
(tests/fault_child.py holds them); here also: the product library refuses the developer keys."""
import os
import subprocess
import sys

import pytest
import torch

from golden_cases import noise

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = os.path.join(ROOT, "audiocodecs_amd", "libaudiocodecs_amd_dev.so")


@pytest.mark.parametrize("case", ["test_failed_persistent_launch_is_reported_and_healed", "test_strict_mode_raises_in_the_call_that_failed",
                                  "test_one_poll_reports_and_clears_every_pending_failure"])
def test_fault_injection_in_the_developer_library(case):
    assert os.path.exists(DEV), "build the developer library (audiocodecs_amd/csrc/build.sh)"
    env = dict(os.environ, AUDIOCODECS_AMD_LIB=DEV)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fault_child.py"), case], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and f"FAULT_CHILD_OK {case}" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_the_product_library_refuses_the_developer_switches(checkpoints):
    from audiocodecs_amd import Encodec
    from audiocodecs_amd._native import NativeError, debug_set

    cfg, sd = checkpoints("full", 0)
    codec = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    sig = noise(8290, 1, 4000).cuda()
    want = codec.sig_to_toks(sig)
    for key in ("lstm_dbg", "rb6_dbg"):
        with pytest.raises(NativeError, match="developer-build switch"):
            debug_set(codec, key, 16)
    assert torch.equal(codec.sig_to_toks(sig), want)
