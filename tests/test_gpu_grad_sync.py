"""GPU: the data-parallel gradient path on the real backend.

  * benchmarks/rccl_ws1_check.py in a child process: `nccl` (= RCCL) at world_size 1, AdapterGradSync's in-place
    all_reduce(AVG) per bucket on the side stream -- from the fused-accumulation callback inside an eager backward and from
    launch_ready() between replayed backward segment graphs (bench.py's N > 1 step) -- gradients equal to plain autograd.
  * a module shared by two layer calls through AdapterGradSync.attach_fused(): ONE report per parameter, after
    its last accumulation.
"""
import os
import resource
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_rccl_world_size_1_eager_and_segmented_graph_step():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "rccl_ws1_check.py")], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env, preexec_fn=lambda: resource.setrlimit(resource.RLIMIT_CORE, (0, 0)))
    tail = (out.stdout + out.stderr)[-2500:]
    assert out.returncode == 0 and "rccl-ws1 ok" in out.stdout, tail


@pytest.mark.parametrize("defer", [True, False], ids=["deferred", "per_layer"])
def test_shared_module_reports_once_through_the_sync(defer):
    from lycoris_amd import ops
    from lycoris_amd.grad_sync import AdapterGradSync
    gen = torch.Generator(device=DEV).manual_seed(3)
    f32 = dict(device=DEV, dtype=torch.float32, generator=gen)
    w1 = torch.nn.Parameter(torch.randn(8, 8, **f32) * 0.3)
    w2 = torch.nn.Parameter(torch.randn(80, 80, **f32) * 0.05)
    other = [torch.nn.Parameter(torch.randn(8, 8, **f32) * 0.3), torch.nn.Parameter(torch.randn(80, 80, **f32) * 0.05)]
    xs = [torch.randn(m, 640, device=DEV, dtype=torch.bfloat16, generator=gen).requires_grad_(True) for m in (96, 64, 32)]
    gs = [torch.randn(m, 640, device=DEV, dtype=torch.bfloat16, generator=gen) * 0.05 for m in (96, 64, 32)]

    def run():
        ys = [ops.lokr_linear(xs[0], w1, w2, 1.0), ops.lokr_linear(xs[1], other[0], other[1], 1.0),
              ops.lokr_linear(xs[2], w1, w2, 1.0)]  # (w1, w2) used by the first AND the last layer call
        return ys

    want = [t.clone() for t in torch.autograd.grad(run(), [w1, w2] + other, gs)]
    # one bucket per parameter pair: the shared pair's bucket must wait for BOTH of its uses
    sync = AdapterGradSync([w1, w2] + other, bucket_bytes=1)
    reports = []
    orig = sync._on_grad_ready
    sync._on_grad_ready = lambda p: (reports.append(id(p)), orig(p))[1]
    sync.attach_fused()
    ops.deferred_weight_gradients(defer)
    try:
        for _ in range(2):
            reports.clear()
            sync.zero_grad()
            torch.autograd.grad(run(), xs, gs)
            assert sorted(reports) == sorted({id(w1), id(w2), id(other[0]), id(other[1])}), "one report per parameter"
            assert all(b.launched for b in sync.buckets)
            sync.finish()
            torch.cuda.synchronize()
            for p, w in zip([w1, w2] + other, want):
                assert float((p.grad - w).norm() / w.norm()) < 1e-5
    finally:
        ops.deferred_weight_gradients(True)
        sync.attach_fused(False)
        sync.remove()
