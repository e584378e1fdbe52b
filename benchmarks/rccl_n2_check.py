#!/usr/bin/env python3
"""The data-parallel adapter-gradient exchange with MORE THAN ONE RCCL rank (VERDICT r5 missing #3 / next #8b).

Until round 6 `ncclCommInitRank(world > 1)` of this code base had never executed anywhere: the builder's boxes have one GPU, and the
driver's 8-GPU lease went straight to bench.py.  This script is what tests/test_gpu_zz_rccl_n2.py launches when >= 2 GPUs are visible:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port P benchmarks/rccl_n2_check.py

Every rank: `RcclCommunicator.from_env` (no process group), the primitives against hand-computed values, then the SAME adapted layers
(same seeded parameters on every rank, rank-seeded activations) through `AdapterGradSync(comm=...)`:
  * one eager step, bucket collectives launched from the kernels' fused-accumulation callback on the communicator's own stream,
  * one captured step (forward graph + backward segment graphs, collectives between the replays) -- bench.py's overlapped form,
  * one captured step with the collectives on the compute stream behind ONE backward graph -- bench.py's inline form,
and compares every gradient with the HAND-AVERAGED single-process gradients (each rank recomputes all ranks' micro-batches locally with
plain autograd and averages them in float64).  Prints "rccl-n2 ok ranks=<ncclCommCount>" on rank 0 and exits 0.
"""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from lycoris_amd import ops
from lycoris_amd.grad_sync import AdapterGradSync, RcclCommunicator

SHAPES = [("lokr", 256, 640, 640), ("locon", 256, 640, 1280), ("lokr", 64, 1280, 640), ("locon", 77, 2048, 640)] * 4


def make_params(dev):
    gen = torch.Generator(device=dev).manual_seed(11)  # the SAME parameters on every rank
    f32 = dict(device=dev, dtype=torch.float32, generator=gen)
    out = []
    for algo, M, I, O in SHAPES:
        if algo == "lokr":
            out.append([torch.nn.Parameter(torch.randn(8, 8, **f32) * 0.3), torch.nn.Parameter(torch.randn(O // 8, I // 8, **f32) * 0.05)])
        else:
            out.append([torch.nn.Parameter(torch.randn(16, I, **f32) * 0.05), torch.nn.Parameter(torch.randn(O, 16, **f32) * 0.05)])
    return out


def make_batch(dev, rank):
    gen = torch.Generator(device=dev).manual_seed(100 + rank)  # every rank its own micro-batch
    xs, gs = [], []
    for algo, M, I, O in SHAPES:
        xs.append(torch.randn(M, I, device=dev, dtype=torch.bfloat16, generator=gen).requires_grad_(True))
        gs.append(torch.randn(M, O, device=dev, dtype=torch.bfloat16, generator=gen) / O ** 0.5)
    return xs, gs


def forward_all(params, xs):
    return [ops.lokr_linear(x, p[0], p[1], 1.0) if a[0] == "lokr" else ops.locon_linear(x, p[0], p[1], 1.0)
            for a, p, x in zip(SHAPES, params, xs)]


def backward_range(outs, params, xs, gs, lo, hi):
    idx = list(range(lo, hi))[::-1]
    torch.autograd.grad([outs[i] for i in idx], [t for i in idx for t in [xs[i]] + params[i]], [gs[i] for i in idx], allow_unused=True)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    # (LYC_N2_ALLOW_WS1=1: the same script as a 1-rank job -- every line but the cross-rank traffic -- for single-GPU boxes)
    assert (world >= 2 or os.environ.get("LYC_N2_ALLOW_WS1")) and torch.cuda.device_count() >= world, (world, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm = RcclCommunicator.from_env(dev)
    assert not dist.is_initialized() and (comm.rank, comm.world) == (rank, world)
    assert comm.count() == world, (comm.count(), world)            # what RCCL itself says
    # ---- primitives against hand-computed values -------------------------------------------------------------------------------
    t = torch.full((1 << 20,), float(rank + 1), device=dev)
    comm.wait_current()
    comm.all_reduce(t, comm.SUM)
    comm.join()
    torch.cuda.synchronize()
    assert torch.equal(t, torch.full_like(t, world * (world + 1) / 2)), float(t[0])
    full = torch.arange(world * 1024, device=dev, dtype=torch.float32) * (rank + 1)
    shard = full[rank * 1024:(rank + 1) * 1024]
    comm.wait_current()
    comm.reduce_scatter(shard, full, comm.SUM)      # in place: my slice of the sum over ranks
    comm.all_gather(full, shard)                    # ... and everyone's slice back
    comm.join()
    torch.cuda.synchronize()
    assert torch.equal(full, torch.arange(world * 1024, device=dev, dtype=torch.float32) * (world * (world + 1) / 2))
    b = torch.full((4096,), float(rank), device=dev)
    comm.wait_current()
    comm.broadcast(b, world - 1)
    comm.join()
    torch.cuda.synchronize()
    assert torch.equal(b, torch.full_like(b, world - 1.0))
    comm.barrier()
    assert comm.max_over_ranks(float(rank)) == world - 1.0
    if rank == 0:
        print(f"rccl communicator up: {comm.count()} ranks (ProcessGroup-free), primitives ok", flush=True)

    # ---- truth: every rank's micro-batch recomputed here with plain autograd, averaged in float64 ----------------------------------
    params = make_params(dev)
    flat = [p for ps in params for p in ps]
    want = [torch.zeros(p.shape, dtype=torch.float64, device=dev) for p in flat]
    for r in range(world):
        xs_r, gs_r = make_batch(dev, r)
        grads = torch.autograd.grad(forward_all(params, xs_r), flat, gs_r)
        for w, g in zip(want, grads):
            w += g.double() / world
    torch.cuda.synchronize()
    xs, gs = make_batch(dev, rank)
    n = len(SHAPES)

    def check(tag):
        torch.cuda.synchronize()
        worst = max(float((p.grad.double() - w).norm() / (w.norm() + 1e-30)) for p, w in zip(flat, want))
        assert worst < 2e-5, f"rank {rank} {tag}: gradient differs from the hand-averaged one by {worst:.3e}"
        return worst

    # ---- eager: collectives from inside the backward (fused-accumulation callback), communicator's own stream ---------------------
    sync = AdapterGradSync(flat, bucket_bytes=256 << 10, always_reduce=True, comm=comm)
    assert len(sync.buckets) >= 3
    sync.attach_fused()
    try:
        for rep in range(2):
            sync.zero_grad()
            outs = forward_all(params, xs)
            backward_range(outs, params, xs, gs, 0, n)
            in_backward = len(sync.launch_log)
            sync.finish()
            e = check(f"eager step {rep}")
        assert in_backward == len(sync.buckets), (in_backward, len(sync.buckets))
        if rank == 0:
            print(f"eager: {len(sync.buckets)} buckets all-reduced (AVG) from inside the backward, rel-err {e:.1e}", flush=True)
        del outs
        # ---- captured, overlapped form: backward segment graphs, collectives between the replays ------------------------------
        torch.cuda.synchronize()
        comm.barrier()
        sync._sync_enabled = False
        pool = torch.cuda.graph_pool_handle()
        g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fwd, pool=pool, capture_error_mode="thread_local"):
            for arena in sync.arenas.values():
                arena.zero_()
            outs = forward_all(params, xs)
        order = {p: i for i, ps in enumerate(params) for p in ps}
        cuts = sync.bucket_boundaries(order)
        edges = sorted(set(cuts), reverse=True)
        plan, done = [], set()
        for e_ in edges:
            ready = [i for i, c in enumerate(cuts) if c >= e_ and i not in done]
            done.update(ready)
            plan.append(ready)
        graphs, hi = [], n
        for e_ in edges:
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, pool=pool, capture_error_mode="thread_local"):
                backward_range(outs, params, xs, gs, e_, hi)
            graphs.append(gph)
            hi = e_
        assert hi == 0
        sync._sync_enabled = True
        for rep in range(3):
            sync._reset_pending()
            g_fwd.replay()
            for gph, ready in zip(graphs, plan):
                gph.replay()
                sync.launch_buckets(ready, after=sync.mark())
            sync.finish()
            e = check(f"captured (overlapped) step {rep}")
        if rank == 0:
            print(f"captured: 1 forward graph + {len(graphs)} bucket-aligned backward segments, collectives between the replays, rel-err {e:.1e}",
                  flush=True)
    finally:
        sync.attach_fused(False)
        sync.remove()
    # ---- captured, inline form: ONE backward graph, every collective on the compute stream behind it (bench.py's small-payload mode) --
    comm2 = RcclCommunicator.from_env(dev, on_current_stream=True)
    assert comm2.count() == world
    for p in flat:
        p.grad = None
    sync = AdapterGradSync(flat, bucket_bytes=256 << 10, always_reduce=True, comm=comm2)
    sync.attach_fused()
    try:
        sync.zero_grad()
        outs = forward_all(params, xs)
        backward_range(outs, params, xs, gs, 0, n)   # warm-up of this arena
        sync.finish()
        check("inline eager step")
        del outs
        torch.cuda.synchronize()
        comm2.barrier()
        sync._sync_enabled = False
        g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fwd, pool=pool, capture_error_mode="thread_local"):
            for arena in sync.arenas.values():
                arena.zero_()
            outs = forward_all(params, xs)
        g_bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_bwd, pool=pool, capture_error_mode="thread_local"):
            backward_range(outs, params, xs, gs, 0, n)
        sync._sync_enabled = True
        for rep in range(3):
            sync._reset_pending()
            g_fwd.replay()
            g_bwd.replay()
            sync.finish()
            e = check(f"captured (inline) step {rep}")
        if rank == 0:
            print(f"captured: ONE backward graph, {len(sync.buckets)} collectives on the compute stream behind it, rel-err {e:.1e}", flush=True)
    finally:
        sync.attach_fused(False)
        sync.remove()
    comm.barrier()
    ranks = comm.count()
    comm2.destroy()
    comm.destroy()
    if rank == 0:
        print(f"rccl-n2 ok ranks={ranks}", flush=True)


if __name__ == "__main__":
    main()
