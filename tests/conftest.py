import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The suites bind the in-tree libslam_engine.so (git-ignored): build it when a fresh checkout has none
    (hipcc cross-compiles gfx950 without a GPU). Never falls back to anything else."""
    lib = os.path.join(ROOT, "slamkit_amd", "lib", "libslam_engine.so")
    if not os.path.exists(lib):
        from slamkit_amd.csrc import build as B
        B.build()


@pytest.fixture(scope="session")
def golden_npz():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "tiny_model.npz")))


@pytest.fixture(scope="session")
def golden_data():
    import json
    return json.load(open(os.path.join(GOLDEN, "data.json")))


def load_wide_golden():
    """tests/golden/wide_model.npz (make_golden_wide.py): head_dim 128, vocab 700, rope_theta 1e6."""
    import numpy as np
    g = dict(np.load(os.path.join(GOLDEN, "wide_model.npz")))
    conv = {"rms_eps": float, "rope_theta": float}
    cfg = {k: conv.get(k, int)(float(v)) for k, v in g["meta_config"]}
    seed, bias_std, jit = g["meta_init"]
    return g, cfg, int(seed), float(bias_std), float(jit)


@pytest.fixture(scope="session")
def wide_golden():
    return load_wide_golden()
