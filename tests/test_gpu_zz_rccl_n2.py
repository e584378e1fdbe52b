"""More than one RCCL rank (VERDICT r5 missing #3 / next #8b): skipped unless >= 2 GPUs are visible, so that a multi-GPU lease runs
`ncclCommInitRank(world > 1)`, the communicator's primitives, one eager and two captured data-parallel steps (overlapped and inline
collectives) and compares the synchronised gradients with the hand-averaged single-process ones BEFORE the scaling bench does.
The work is benchmarks/rccl_n2_check.py under the launcher the driver uses (python -m torch.distributed.run on 127.0.0.1).
(File name: last in the alphabetical order of the suite -- it is the only test that needs hardware the builder never had.)"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(nproc, extra_env=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "benchmarks", "rccl_n2_check.py")]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)


def test_the_check_script_as_a_one_rank_job_under_the_launcher():
    """every line of the script except the cross-rank traffic, on any box: from_env under torch.distributed.run with WORLD_SIZE=1"""
    out = _launch(1, {"LYC_N2_ALLOW_WS1": "1"})
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0 and "rccl-n2 ok ranks=1" in out.stdout, tail


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (one process per GPU)")
def test_two_rccl_ranks_average_the_adapter_gradients_eager_and_captured():
    out = _launch(2)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0 and "rccl-n2 ok ranks=2" in out.stdout, tail


@pytest.mark.skipif(torch.cuda.device_count() < 8, reason="needs the 8-GPU node")
def test_eight_rccl_ranks_average_the_adapter_gradients_eager_and_captured():
    out = _launch(8)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0 and "rccl-n2 ok ranks=8" in out.stdout, tail
