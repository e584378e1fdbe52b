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


# test files whose FROZEN layers are tiny fp32 convolutions (module-level golden cases: the fault below was seen in exactly these)
_OFF_MIOPEN = ("test_gpu_modules_golden.py", "test_gpu_golden_sweep.py")


@pytest.fixture(autouse=True)
def _frozen_convs_off_miopen(request):
    """The FROZEN fp32 convolutions of the module-level golden tests (not this repository's code: DESIGN 0, "the frozen layers' own GEMMs /
    convolutions stay with rocBLAS / MIOpen") run through ATen's own convolution instead of MIOpen.  MIOpen's Find step for the
    backward-data of these tiny fp32 layers page-faults on SOME boxes of the pool -- deterministic per box, absent on others, the faulting
    dispatch a Tensile SGEMM of the library with no kernel of this repository in flight (profiles/
    r06_c1_five_file_blocking_*.log, r05_abort_runA.log; seen again in round 6's closing run inside tests/test_gpu_golden_sweep.py,
    profiles/r06_final_pytest_gpu_abort.log) -- and takes the whole pytest process with it.  The adapter path under test is unaffected (it
    never calls MIOpen); every other test file, bench.py and the product keep MIOpen for the frozen layers (16-bit frozen convolutions are
    also rounded differently by ATen's own path: the autocast tests' bounds are MIOpen's).  LYC_TEST_MIOPEN=1 keeps MIOpen on here too."""
    if os.path.basename(str(request.node.fspath)) not in _OFF_MIOPEN or os.environ.get("LYC_TEST_MIOPEN") == "1":
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    prev = torch.backends.cudnn.enabled
    torch.backends.cudnn.enabled = False
    try:
        yield
    finally:
        torch.backends.cudnn.enabled = prev


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
