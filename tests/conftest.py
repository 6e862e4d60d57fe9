import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "rfdata_golden.npz")
    return np.load(path)


@pytest.fixture(scope="session")
def golden_all():
    """Every full-size capture of the reference's rfdata/ (35 files): first 2*8192 bytes, rows at 1024 and 8192, and one
    capture's whole 262144-byte block with its 128 rows at 1024 (tests/golden/make_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "rfdata_all_golden.npz"))


def _all_capture_keys():
    with np.load(os.path.join(ROOT, "tests", "golden", "rfdata_all_golden.npz")) as z:
        return sorted(k[:-len("__sha256")] for k in z.files if k.endswith("__sha256"))


ALL_CAPTURE_KEYS = _all_capture_keys()
GOLDEN_KEYS = ["rf_100p900_1", "rf_202p500_1", "rf_202p500_2", "rf_202p500_3"]
GOLDEN_SIZES = [128, 256, 1024, 4096, 8192, 16384]


def synth_iq(seed, n_bytes, tone_bin_frac=0.125, sigma=20.0, amp=40.0):
    """HackRF-style int8 IQ bytes: Gaussian sigma=20 plus one
    complex tone at +fs*tone_bin_frac, amplitude 40, stored as raw int8 bytes."""
    rng = np.random.default_rng(seed)
    n = n_bytes // 2
    t = np.arange(n)
    tone = amp * np.exp(2j * np.pi * tone_bin_frac * t)
    i = rng.normal(0, sigma, n) + tone.real
    q = rng.normal(0, sigma, n) + tone.imag
    iq = np.empty(2 * n, dtype=np.float64)
    iq[0::2] = i
    iq[1::2] = q
    return np.clip(np.rint(iq), -128, 127).astype(np.int8).view(np.uint8)


def kernel_stem(n, mode):
    """Symbol stem of the kernels a plan of (fft_size, mode) launches: where the modes of a size prefer different radix
    orders the plan takes its mode's configuration (frequensea_amd/csrc/fsea_configs.h, fsea_api.hip: preferred_variant)."""
    if n == 256 and mode in (0, 4, 5):
        return "fsea_fft256rows"
    if n == 512 and mode in (1, 2):
        return "fsea_fft512px"
    if n == 1024 and mode in (3, 4, 5):
        return "fsea_fft1024rt"
    return "fsea_fft%d" % n
