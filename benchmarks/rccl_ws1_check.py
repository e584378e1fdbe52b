#!/usr/bin/env python3
"""The data-parallel gradient path through RCCL on ONE GPU (missing #5 / next #8).

No 8-GPU node is available to the builder, so until the driver's first N = 8 run `AdapterGradSync`'s collectives had only ever
run through gloo.  This script initialises the `nccl` (= RCCL) backend at world_size 1 and drives the SAME code the N > 1 bench
step runs -- in-place `all_reduce(ReduceOp.AVG)` of each bucket on the side stream, launched from the kernels' fused-accumulation
callback (eager) and by `launch_ready()` between the replays of the backward segment graphs (captured), `finish()` joining the
side stream before the optimizer -- and checks that the gradients equal the ones of a plain single-process backward.

    MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 WORLD_SIZE=1 python benchmarks/rccl_ws1_check.py

Round 5: `--comm rccl` runs the same steps through the ProcessGroup-free communicator (lycoris_amd.grad_sync.RcclCommunicator:
ncclCommInitRank once, collectives as plain work on its own HIP stream) -- no torch.distributed process group exists -- plus the
communicator's primitives on their own and a step whose bucket collectives are RECORDED INSIDE the backward hipGraph.

Also checks a module shared by two layer calls through the sync (its bucket used to be reduced early).
Prints "rccl-ws1 ok" and exits 0.  Run by tests/test_gpu_grad_sync.py in a child process.
"""
import faulthandler
import os
import sys
import time

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from lycoris_amd import ops
from lycoris_amd.grad_sync import AdapterGradSync, RcclCommunicator

DEV = torch.device("cuda:0")


class Layer:
    def __init__(self, algo, M, I, O, gen, params=None):
        self.algo = algo
        self.x = torch.randn(M, I, device=DEV, dtype=torch.bfloat16, generator=gen).requires_grad_(True)
        self.g = torch.randn(M, O, device=DEV, dtype=torch.bfloat16, generator=gen) / O ** 0.5
        f32 = dict(device=DEV, dtype=torch.float32, generator=gen)
        if params is not None:
            self.params = params
        elif algo == "lokr":
            self.params = [torch.nn.Parameter(torch.randn(8, 8, **f32) * 0.3), torch.nn.Parameter(torch.randn(O // 8, I // 8, **f32) * 0.05)]
        else:
            self.params = [torch.nn.Parameter(torch.randn(16, I, **f32) * 0.05), torch.nn.Parameter(torch.randn(O, 16, **f32) * 0.05)]

    def forward(self):
        if self.algo == "lokr":
            return ops.lokr_linear(self.x, self.params[0], self.params[1], 1.0)
        return ops.locon_linear(self.x, self.params[0], self.params[1], 1.0)


def backward_range(outs, lo, hi):
    seg = outs[lo:hi][::-1]
    # the factors are listed: a backward call computes (and the kernels add into `.grad`) only what it is asked for
    torch.autograd.grad([y for y, _ in seg], [t for _, l in seg for t in [l.x] + list(l.params)], [l.g for _, l in seg], allow_unused=True)


def main():
    # diagnostics: --no-pg (no process group at all: the capture logic alone), --skip-eager (captured phase only),
    # --eager-only (what tests/test_gpu_grad_sync.py runs by default: see its docstring)
    no_pg, skip_eager, eager_only = "--no-pg" in sys.argv, "--skip-eager" in sys.argv, "--eager-only" in sys.argv
    torch.cuda.set_device(0)
    own = "--comm" in sys.argv and sys.argv[sys.argv.index("--comm") + 1] == "rccl"
    comm = None
    if own:
        no_pg = True
        comm = RcclCommunicator(0, 1, DEV)
        assert not dist.is_initialized()
        # the primitives, in place, ordered against the compute stream by events only
        t = torch.randn(1 << 20, device=DEV)
        ref = t.clone()
        comm.wait_current()
        comm.all_reduce(t, comm.AVG)
        comm.reduce_scatter(t[:t.numel()], t, comm.SUM)   # world 1: the shard IS the buffer
        comm.all_gather(t, t[:t.numel()])
        comm.broadcast(t, 0)
        comm.join()
        t.add_(1.0)  # on the compute stream, behind the join
        torch.cuda.synchronize()
        assert torch.equal(t, ref + 1.0)
        comm.barrier()
        assert comm.max_over_ranks(3.5) == 3.5
        print("rccl communicator up (ProcessGroup-free), primitives ok", flush=True)
    if not no_pg:
        dist.init_process_group("nccl", device_id=DEV)
        assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
        print("nccl process group up", flush=True)
    gen = torch.Generator(device=DEV).manual_seed(7)
    shapes = [("lokr", 256, 640, 640), ("locon", 256, 640, 1280), ("lokr", 64, 1280, 640), ("locon", 77, 2048, 640)] * 6
    if "--lokr-only" in sys.argv:
        shapes = [s for s in shapes if s[0] == "lokr"]
    if "--locon-only" in sys.argv:
        shapes = [s for s in shapes if s[0] == "locon"]
    layers = [Layer(a, M, I, O, gen) for a, M, I, O in shapes]
    if "--no-shared" not in sys.argv and "--locon-only" not in sys.argv:
        shared = Layer("lokr", 128, 640, 640, gen)
        layers.insert(5, shared)
        layers.append(Layer("lokr", 32, 640, 640, gen, params=shared.params))  # the same parameters in a second layer call
    params, seen = [], set()
    for l in layers:
        for p in l.params:
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)

    # ---- truth: plain autograd, gradients handed back ------------------------------------------------------------------
    # diagnostics for the capture crash (round 3): --side-stream runs everything before the capture on a side stream (as bench.py's
    # warm-up step does), --no-truth skips this phase (the captured step is then compared with nothing: crash test only),
    # --big-buckets uses bench.py's 32 MiB buckets
    side_ctx = torch.cuda.stream(torch.cuda.Stream()) if "--side-stream" in sys.argv else None
    if side_ctx is not None:
        side_ctx.__enter__()
    no_truth = "--no-truth" in sys.argv
    if not no_truth:
        outs = [(l.forward(), l) for l in layers]
        want = torch.autograd.grad([y for y, _ in outs], params, [l.g for _, l in outs])
        want = [w.clone() for w in want]
        del outs
        torch.cuda.synchronize()
        print("truth computed", flush=True)

    def check(tag):
        torch.cuda.synchronize()
        worst = 0.0
        if no_truth:
            return worst
        for p, w in zip(params, want):
            e = float((p.grad - w).norm() / (w.norm() + 1e-30))
            worst = max(worst, e)
        assert worst < 1e-5, f"{tag}: gradient mismatch {worst:.3e}"
        return worst

    # ---- eager: collectives launched from inside the backward by the fused-accumulation callback -------------------------
    sync = AdapterGradSync(params, bucket_bytes=(32 << 20) if "--big-buckets" in sys.argv else (256 << 10), always_reduce=(not no_pg) or own, comm=comm)
    assert len(sync.buckets) >= 3 or len(shapes) < 24 or "--big-buckets" in sys.argv, len(sync.buckets)  # several buckets
    print(f"{len(sync.buckets)} buckets", flush=True)
    sync.attach_fused()
    try:
        for rep in range(0 if skip_eager else 2):
            sync.zero_grad()
            outs = [(l.forward(), l) for l in layers]
            backward_range(outs, 0, len(layers))
            launched_in_backward = len(sync.launch_log)
            sync.finish()
            e = check(f"eager step {rep}")
        if not skip_eager:
            assert launched_in_backward == len(sync.buckets), (launched_in_backward, len(sync.buckets))
            print(f"eager: {len(sync.buckets)} buckets all-reduced (AVG, side stream) from inside the backward, rel-err {e:.1e}", flush=True)

        if eager_only:
            print("rccl-ws1 ok (eager only)", flush=True)
            return
        # ---- captured: forward graph + backward segment graphs, launch_ready() between the replays (bench.py's N > 1 step) ---
        if side_ctx is not None:
            side_ctx.__exit__(None, None, None)
        torch.cuda.synchronize()
        outs = None
        time.sleep(1.0)  # every collective of the eager steps has been retired by RCCL's watchdog thread before the capture starts
        # exactly bench.py's capture sequence (main(), `if not args.eager`): captures on the ambient stream context, one pool
        sync._sync_enabled = False
        pool = torch.cuda.graph_pool_handle()
        g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fwd, pool=pool, capture_error_mode="thread_local"):
            for arena in sync.arenas.values():
                arena.zero_()
            outs = [(l.forward(), l) for l in layers]
        n, nseg = len(layers), 4
        edges = [round(i * n / nseg) for i in range(nseg + 1)]
        graphs, bounds = [], []
        for s in range(nseg, 0, -1):
            lo, hi = edges[s - 1], edges[s]
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, pool=pool, capture_error_mode="thread_local"):
                backward_range(outs, lo, hi)
            graphs.append(gph)
            bounds.append(layers[lo - 1].params[-1] if lo > 0 else None)
        print("captured", flush=True)
        sync._sync_enabled = True
        for rep in range(3):
            sync._reset_pending()
            g_fwd.replay()
            for gph, upto in zip(graphs, bounds):
                gph.replay()
                sync.launch_ready(upto)
            sync.finish()
            e = check(f"graph step {rep}")
        print(f"captured: 1 forward graph + {nseg} backward segments, bucket all-reduces between the replays, rel-err {e:.1e}", flush=True)
        if own:  # the collectives recorded INTO one backward graph: the communicator's stream is forked into the capture and joined back
            cuts = sync.bucket_boundaries({p: i for i, l in enumerate(layers) for p in l.params})
            bedges = sorted(set(cuts), reverse=True)
            done, bplan = set(), []
            for e_ in bedges:
                ready = [i for i, c in enumerate(cuts) if c >= e_ and i not in done]
                done.update(ready)
                bplan.append(ready)
            sync._sync_enabled = False
            sync._reset_pending()
            g_fwd = torch.cuda.CUDAGraph()  # a fresh forward: the segment captures above consumed the first one's autograd graph
            with torch.cuda.graph(g_fwd, pool=pool, capture_error_mode="thread_local"):
                for arena in sync.arenas.values():
                    arena.zero_()
                outs = [(l.forward(), l) for l in layers]
            g_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_bwd, pool=pool, capture_error_mode="thread_local"):
                hi = n
                for e_, ready in zip(bedges, bplan):
                    backward_range(outs, e_, hi)
                    sync.launch_buckets(ready)
                    hi = e_
                if hi > 0:
                    backward_range(outs, 0, hi)
                comm.join()
            sync._comm_pending = False
            sync._sync_enabled = True
            for rep in range(3):
                sync._reset_pending()
                g_fwd.replay()
                g_bwd.replay()
                e = check(f"captured-collectives step {rep}")
            print(f"captured: ONE backward graph with {sum(len(r) for r in bplan)} bucket collectives recorded inside it, rel-err {e:.1e}", flush=True)
    finally:
        sync.attach_fused(False)
        sync.remove()
        if not no_pg and dist.is_initialized():
            dist.destroy_process_group()
        if comm is not None:
            comm.destroy()
    print("rccl-ws1 ok", flush=True)


if __name__ == "__main__":
    main()
