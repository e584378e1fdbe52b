import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the round-end driver)")


@pytest.fixture(scope="session", autouse=True)
def _locon_kernel_choice():
    """LYC_TEST_LOCON_REG=1 runs the whole session with the rank-r launches on the register-staged kernel of rounds 1-5 (the A/B leg of
    the GPU suite: ops.locon_reg_staged); the default is the LDS-DMA kernel of round 6."""
    if os.environ.get("LYC_TEST_LOCON_REG") == "1":
        import torch
        if torch.cuda.is_available():
            from lycoris_amd import ops
            ops.locon_reg_staged(True)
    yield


@pytest.fixture(scope="session")
def golden_cases():
    """name -> (meta dict, {array name -> float64 ndarray}) produced by tests/golden/make_golden.py."""
    blob = np.load(os.path.join(GOLDEN, "adapter_cases.npz"))
    with open(os.path.join(GOLDEN, "adapter_cases.json")) as f:
        metas = json.load(f)
    out = {}
    for name, meta in metas.items():
        arrs = {k.split("/", 1)[1]: blob[k] for k in blob.files if k.startswith(name + "/")}
        out[name] = (meta, arrs)
    return out


def golden_case_names():
    with open(os.path.join(GOLDEN, "adapter_cases.json")) as f:
        return sorted(json.load(f).keys())
