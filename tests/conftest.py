import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    import parity_record

    line = parity_record.write()
    if line:
        terminalreporter.write_line(line)


@pytest.fixture(scope="session")
def golden():
    z = np.load(os.path.join(GOLDEN_DIR, "encodec_golden.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


_CKPT = {}


@pytest.fixture(scope="session")
def checkpoints():
    """(cfg_name, seed) -> (cfg, HF-format synthetic state dict); cached for the session."""
    from audiocodecs_amd import checkpoint
    from audiocodecs_amd.config import ENCODEC_24KHZ, TINY

    def get(cfg_name, seed):
        key = (cfg_name, seed)
        if key not in _CKPT:
            cfg = {"full": ENCODEC_24KHZ, "tiny": TINY}[cfg_name]
            _CKPT[key] = (cfg, checkpoint.synthetic_state_dict(cfg, seed=seed))
        return _CKPT[key]

    return get


@pytest.fixture(scope="session")
def mimi_golden():
    z = np.load(os.path.join(GOLDEN_DIR, "mimi_golden.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


_MIMI_CKPT = {}


@pytest.fixture(scope="session")
def mimi_checkpoints():
    """(cfg_name, seed) -> (MimiConfig, HF-format synthetic Mimi state dict); cached for the session."""
    from audiocodecs_amd import checkpoint
    from audiocodecs_amd.config import MIMI_24KHZ, MIMI_TINY

    def get(cfg_name, seed):
        key = (cfg_name, seed)
        if key not in _MIMI_CKPT:
            cfg = {"full": MIMI_24KHZ, "tiny": MIMI_TINY}[cfg_name]
            _MIMI_CKPT[key] = (cfg, checkpoint.synthetic_mimi_state_dict(cfg, seed=seed))
        return _MIMI_CKPT[key]

    return get


@pytest.fixture(scope="session")
def dac_golden():
    z = np.load(os.path.join(GOLDEN_DIR, "dac_golden.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


_DAC_CKPT = {}


@pytest.fixture(scope="session")
def dac_checkpoints():
    """(cfg_name, seed) -> (DacConfig, HF-layout synthetic DAC state dict); cached for the session."""
    from audiocodecs_amd import checkpoint
    from audiocodecs_amd.config import DAC_44KHZ, DAC_TINY

    def get(cfg_name, seed):
        key = (cfg_name, seed)
        if key not in _DAC_CKPT:
            cfg = {"full": DAC_44KHZ, "tiny": DAC_TINY}[cfg_name]
            _DAC_CKPT[key] = (cfg, checkpoint.synthetic_dac_state_dict(cfg, seed=seed))
        return _DAC_CKPT[key]

    return get


@pytest.fixture(scope="session")
def wavtok_golden():
    z = np.load(os.path.join(GOLDEN_DIR, "wavtokenizer_golden.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


_WT_CKPT = {}


@pytest.fixture(scope="session")
def wavtok_checkpoints():
    """(cfg_name, seed) -> (WavTokenizerConfig, synthetic checkpoint in the upstream key layout); cached for the session."""
    from audiocodecs_amd import checkpoint
    from audiocodecs_amd.config import WAVTOK_40, WAVTOK_75, WAVTOK_TINY

    def get(cfg_name, seed):
        key = (cfg_name, seed)
        if key not in _WT_CKPT:
            cfg = {"full": WAVTOK_40, "f75": WAVTOK_75, "tiny": WAVTOK_TINY}[cfg_name]
            _WT_CKPT[key] = (cfg, checkpoint.synthetic_wavtok_state_dict(cfg, seed=seed))
        return _WT_CKPT[key]

    return get
