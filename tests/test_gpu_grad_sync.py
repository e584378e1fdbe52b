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


def test_processgroup_free_rccl_communicator_world_size_1():
    """round 5: the same steps through lycoris_amd.grad_sync.RcclCommunicator (ncclCommInitRank, own stream, events): primitives,
    eager step with in-backward collectives, segmented graph step, and the collectives recorded inside one backward hipGraph"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "rccl_ws1_check.py"), "--comm", "rccl"], capture_output=True,
                         text=True, timeout=300, cwd=ROOT, env=env, preexec_fn=lambda: resource.setrlimit(resource.RLIMIT_CORE, (0, 0)))
    tail = (out.stdout + out.stderr)[-2500:]
    assert out.returncode == 0 and "rccl-ws1 ok" in out.stdout and "recorded inside it" in out.stdout, tail


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
            torch.autograd.grad(run(), xs + [w1, w2] + other, gs, allow_unused=True)  # the factors are asked for; the kernels add them in place
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


def test_loss_backward_counts_every_fused_parameter_once():
    """loss.backward() runs every leaf's AccumulateGrad node -- and AdapterGradSync's post-accumulate hook -- even for parameters
    whose backward node returned no gradient because the kernels accumulated into `.grad` themselves.  Each parameter must be
    counted ONCE (by the kernels' report), the bucket launched after the LAST of them, two micro-batches under no_sync()."""
    import torch.nn as nn
    from lycoris_amd import ops
    from lycoris_amd.grad_sync import AdapterGradSync
    from lycoris_amd.modules import IA3Module, LoConModule, LokrModule
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    l1, l2, l3 = (nn.Linear(128, 128).to(dev, torch.bfloat16).requires_grad_(False) for _ in range(3))
    mods = [LoConModule("a", l1, 1.0, lora_dim=8, alpha=4), LokrModule("b", l2, 1.0, lora_dim=100000, alpha=1, factor=8),
            IA3Module("c", l3, 1.0)]
    for m in mods:
        m.to(dev)
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn_like(p) * 0.1)
        m.apply_to()
    params = [p for m in mods for p in m.parameters()]
    sync = AdapterGradSync(params, bucket_bytes=1 << 30)  # ONE bucket: it may launch only after the last report
    sync.attach_fused()
    seen = []
    orig = sync._on_grad_ready
    sync._on_grad_ready = lambda p: (seen.append(id(p)), orig(p))[1]
    ops.fused_grad_accumulation(True, callback=sync._on_grad_ready, batch_callback=lambda ps: [sync._on_grad_ready(p) for p in ps])
    try:
        sync.zero_grad()
        for mb in range(2):
            x = torch.randn(64, 128, device=dev).to(torch.bfloat16).requires_grad_(True)
            y = l3(l2(l1(x)))
            if mb == 0:
                with sync.no_sync():
                    y.backward(torch.ones_like(y))
            else:
                y.backward(torch.ones_like(y))  # raised "bucket already all-reduced" before the hook was made aware of the reports
        assert sync.launch_log == [0]
        sync.finish()
        torch.cuda.synchronize()
    finally:
        for m in mods:
            m.restore()
        ops.discard_deferred()
        sync.attach_fused(False)
        sync.remove()
    # both micro-batches reached the hook / callback; each parameter exactly once per micro-batch
    assert sorted(seen) == sorted([id(p) for p in params] * 2)
    for p in params:
        assert p.grad is not None and float(p.grad.abs().max()) > 0


@pytest.mark.parametrize("reentrant", [False, True], ids=["non_reentrant", "reentrant"])
def test_activation_checkpointing_keeps_the_in_backward_reports(reentrant):
    """under torch.utils.checkpoint the forward runs twice (original + recomputation) but only one set of backward
    nodes runs.  Every parameter must still be reported from INSIDE the backward pass (all buckets launched before finish()), once,
    with the right gradients -- including a module applied twice."""
    import torch.nn as nn
    from torch.utils.checkpoint import checkpoint
    from lycoris_amd import ops
    from lycoris_amd.grad_sync import AdapterGradSync
    gen = torch.Generator(device=DEV).manual_seed(5)
    f32 = dict(device=DEV, dtype=torch.float32, generator=gen)
    pairs = [(nn.Parameter(torch.randn(8, 8, **f32) * 0.3), nn.Parameter(torch.randn(80, 80, **f32) * 0.05)) for _ in range(3)]
    params = [p for pr in pairs for p in pr]
    x = torch.randn(128, 640, device=DEV, dtype=torch.bfloat16, generator=gen).requires_grad_(True)
    g = torch.randn(128, 640, device=DEV, dtype=torch.bfloat16, generator=gen) * 0.05

    def block(h):
        for k in (0, 1, 0):  # pair 0 is applied twice inside the checkpointed region
            h = h + ops.lokr_linear(h, pairs[k][0], pairs[k][1], 1.0)
        return h

    def net(h, ckpt):
        h = checkpoint(block, h, use_reentrant=reentrant) if ckpt else block(h)
        return h + ops.lokr_linear(h, pairs[2][0], pairs[2][1], 1.0)

    want = [t.clone() for t in torch.autograd.grad(net(x, False), params, g)]
    sync = AdapterGradSync(params, bucket_bytes=1)
    sync.attach_fused()
    try:
        for _ in range(2):
            sync.zero_grad()
            net(x, True).backward(g)
            assert sorted(sync.launch_log) == list(range(len(sync.buckets))), "every bucket launched from inside the backward pass"
            sync.finish()
            torch.cuda.synchronize()
            for p, w in zip(params, want):
                assert float((p.grad - w).norm() / w.norm()) < 1e-5
    finally:
        sync.attach_fused(False)
        sync.remove()
