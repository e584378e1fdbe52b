"""N > 1 path of the adapter-gradient sync, covered with world_size-2 gloo processes on CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, bucket_bytes, out, collective="all_reduce"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lycoris_amd.grad_sync import AdapterGradSync
        torch.manual_seed(0)  # identical replicas
        params = [torch.nn.Parameter(torch.randn(s)) for s in [(8, 8), (160, 16), (3,), (40, 5), ()]]
        sync = AdapterGradSync(params, bucket_bytes=bucket_bytes, collective=collective)
        assert sync.world_size == world
        results = []
        for step in range(2):
            sync.zero_grad()
            torch.manual_seed(100 * step + rank)  # different data per rank
            noise = [torch.randn_like(p) for p in params]
            # gradient accumulation: the first micro-batch only accumulates (no_sync), the second one reduces
            with sync.no_sync():
                sum((p ** 2).sum() for p in params[:3]).backward()
                assert sync.launch_log == []
            loss = sum((p * n).sum() for p, n in zip(params, noise))
            loss.backward()
            assert sorted(sync.launch_log) == list(range(len(sync.buckets)))  # every bucket went out during backward
            sync.finish()
            results.append([p.grad.clone() for p in params])
            # every .grad is still a view of the arena
            for p in params:
                assert p.grad.untyped_storage().data_ptr() == sync.arenas[p.dtype].untyped_storage().data_ptr()
        # reference: same computation, gradients averaged by hand
        want = []
        for step in range(2):
            acc = None
            for r in range(world):
                torch.manual_seed(100 * step + r)
                gs = [torch.randn_like(p) for p in params]
                gs = [g + (2 * p.detach() if i < 3 else 0) for i, (g, p) in enumerate(zip(gs, params))]
                acc = gs if acc is None else [a + g for a, g in zip(acc, gs)]
            want.append([a / world for a in acc])
        ok = all(torch.allclose(g, w, atol=1e-6) for gs, ws in zip(results, want) for g, w in zip(gs, ws))
        out.put((rank, ok, len(sync.buckets), sync.payload_bytes))
    finally:
        dist.destroy_process_group()


def _worker_direct(rank, world, port, out):
    """bench.py's pattern: gradients written straight into the arena (no autograd hooks), then all_reduce_now()."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lycoris_amd.grad_sync import AdapterGradSync
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(s)) for s in [(8, 8), (160, 16), (3,)]]
        sync = AdapterGradSync(params, bucket_bytes=256)
        ok = True
        for step in range(3):  # no zero_grad() between steps 1 and 2: finish() must re-arm the buckets itself
            if step != 2:
                sync.zero_grad()
            for p in params:
                p.grad.add_(float(rank + 1 + step))  # "kernel" accumulating into the arena
            sync.all_reduce_now()
            sync.finish()
            want = sum(r + 1 + step for r in range(world)) / world
            if step == 2:  # accumulated on top of the averaged step-1 values, which are equal on every rank
                want = want + sum(r + 1 + 1 for r in range(world)) / world
            ok = ok and all(torch.allclose(p.grad, torch.full_like(p, want)) for p in params)
        out.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_direct_arena_writes_then_all_reduce_now():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_direct, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(2))
    assert [g[1] for g in got] == [True, True], got


@pytest.mark.parametrize("bucket_bytes", [1 << 30, 1024, 1])
def test_gradients_are_averaged_across_two_ranks(bucket_bytes):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(2))
    assert [g[1] for g in got] == [True, True], got
    n_buckets = got[0][2]
    assert n_buckets == (1 if bucket_bytes == 1 << 30 else (5 if bucket_bytes == 1 else n_buckets))
    assert got[0][3] == 4 * (64 + 2560 + 3 + 200 + 1)


@pytest.mark.parametrize("bucket_bytes", [1 << 30, 1024, 1])
def test_reduce_scatter_all_gather_buckets_average_like_all_reduce(bucket_bytes):
    """collective="reduce_scatter" (VERDICT r3 #8): every bucket as an in-place reduce-scatter + all-gather pair.  The bucket sizes
    here are odd (2828 / 2624 + ... / 1 elements over 2 ranks): the shard arithmetic and the all-reduced tail are both exercised."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, out, "reduce_scatter")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(2))
    assert [g[1] for g in got] == [True, True], got


def _worker_segments(rank, world, port, out):
    """bench.py's captured-step pattern: the backward pass replayed in segments cut at the bucket boundaries, the buckets of each
    segment launched by index (no hooks, no scan)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lycoris_amd.grad_sync import AdapterGradSync
        torch.manual_seed(0)
        layers = [[torch.nn.Parameter(torch.randn(8, 8)), torch.nn.Parameter(torch.randn(16, 4))] for _ in range(7)]
        params = [p for l in layers for p in l]
        sync = AdapterGradSync(params, bucket_bytes=600, collective="reduce_scatter")
        order = {p: i for i, l in enumerate(layers) for p in l}
        cuts = sync.bucket_boundaries(order)
        assert cuts == sorted(cuts, reverse=True) and cuts[-1] == 0 and len(cuts) == len(sync.buckets) > 2
        # one segment per DISTINCT boundary, from the last layer down; the buckets completed by each
        edges = sorted(set(cuts), reverse=True)
        plan = [[i for i, c in enumerate(cuts) if c == e] for e in edges]
        sync.zero_grad()
        hi = len(layers)
        for e, buckets in zip(edges, plan):
            for l in layers[e:hi][::-1]:  # "replay" of the segment: the kernels write into the arena
                for p in l:
                    p.grad.add_(float(rank + 1))
            sync.launch_buckets(buckets)
            hi = e
        assert sorted(sync.launch_log) == list(range(len(sync.buckets)))
        sync.finish()
        want = sum(r + 1 for r in range(world)) / world
        out.put((rank, all(torch.allclose(p.grad, torch.full_like(p, want)) for p in params), len(edges)))
    finally:
        dist.destroy_process_group()


def test_bucket_aligned_segments_launch_every_bucket_once():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_segments, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(2))
    assert [g[1] for g in got] == [True, True], got


def test_single_process_arena_semantics():
    from lycoris_amd.grad_sync import AdapterGradSync
    params = [torch.nn.Parameter(torch.randn(4, 4)), torch.nn.Parameter(torch.randn(7))]
    sync = AdapterGradSync(params)
    assert sync.world_size == 1 and len(sync.buckets) == 1
    (params[0].sum() * 2 + params[1].sum()).backward()
    sync.finish()
    assert torch.all(params[0].grad == 2) and torch.all(params[1].grad == 1)
    arena = sync.arenas[torch.float32]
    assert float(arena.sum()) == 2 * 16 + 7
    sync.zero_grad()
    assert float(arena.abs().sum()) == 0 and float(params[0].grad.abs().sum()) == 0
    params[0].grad = None  # e.g. optimizer.zero_grad(set_to_none=True)
    sync.zero_grad()
    assert params[0].grad is not None and params[0].grad.untyped_storage().data_ptr() == arena.untyped_storage().data_ptr()
    sync.remove()


def test_unguarded_second_backward_raises():
    """a second backward after the buckets were reduced must not silently diverge the replicas."""
    from lycoris_amd.grad_sync import AdapterGradSync
    params = [torch.nn.Parameter(torch.randn(4, 4)), torch.nn.Parameter(torch.randn(7))]
    sync = AdapterGradSync(params, bucket_bytes=1)
    sync.zero_grad()
    (params[0].sum() + params[1].sum()).backward()
    with pytest.raises(RuntimeError, match="no_sync"):
        (params[0].sum() + params[1].sum()).backward()
    sync.finish()
    sync.zero_grad()
    with sync.no_sync():
        (params[0].sum() + params[1].sum()).backward()
    (params[0].sum() + params[1].sum()).backward()  # fine: the first one did not count
    sync.finish()
    assert torch.all(params[0].grad == 2)
    sync.remove()


def test_bucket_collective_is_issued_before_the_first_layers_backward():
    """Overlap by construction: with one bucket per layer, the output-side layer's bucket is launched while the
    input-side layer's backward has not run yet (recorded through a tensor hook on the first layer's output)."""
    from lycoris_amd.grad_sync import AdapterGradSync
    l1, l2 = torch.nn.Linear(6, 6), torch.nn.Linear(6, 3)
    sync = AdapterGradSync(list(l1.parameters()) + list(l2.parameters()), bucket_bytes=1)
    events = []
    orig = sync._launch
    sync._launch = lambda b: (events.append(("launch", id(b.params[0]))), orig(b))[1]
    h = l1(torch.randn(5, 6))
    h.register_hook(lambda g: events.append(("l1_backward_starts", None)))
    l2(h).sum().backward()
    first_l1 = next(i for i, e in enumerate(events) if e[0] == "l1_backward_starts")
    l2_ids = {id(p) for p in l2.parameters()}
    launched_before = {e[1] for e in events[:first_l1] if e[0] == "launch"}
    assert launched_before and launched_before <= l2_ids, events
    sync.finish()
    sync.remove()


def test_fused_accumulation_callback_drives_the_buckets():
    """ops.fused_grad_accumulation(callback=...) is what the kernels call instead of the autograd hooks."""
    from lycoris_amd import ops
    from lycoris_amd.grad_sync import AdapterGradSync
    params = [torch.nn.Parameter(torch.randn(4, 4)), torch.nn.Parameter(torch.randn(7))]
    sync = AdapterGradSync(params, bucket_bytes=1)
    sync.attach_fused()
    try:
        assert ops._ACCUM["enabled"] and ops._ACCUM["callback"] == sync._on_grad_ready
        sync.zero_grad()
        ops._ACCUM["callback"](params[1])  # what _finish_grads() does after a kernel accumulated into p.grad
        assert sync.launch_log == [0]      # reverse registration order: params[1] sits in bucket 0
        ops._ACCUM["callback"](params[0])
        assert sync.launch_log == [0, 1]
        sync.finish()
    finally:
        sync.attach_fused(False)
        sync.remove()
    assert not ops._ACCUM["enabled"] and ops._ACCUM["callback"] is None


def _worker_ws1(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lycoris_amd.grad_sync import AdapterGradSync
    params = [torch.nn.Parameter(torch.ones(4, 4)), torch.nn.Parameter(torch.ones(7))]
    plain = AdapterGradSync(params, bucket_bytes=1)
    assert not plain._reduce  # world_size 1: no collective unless asked for
    plain.remove()
    sync = AdapterGradSync(params, bucket_bytes=1, always_reduce=True)
    assert sync._reduce and sync.world_size == 1
    sync.zero_grad()
    (params[0].sum() * 2 + params[1].sum()).backward()
    launched = sync.collectives_launched
    sync.finish()
    ok = launched == 2 and bool(torch.all(params[0].grad == 2)) and bool(torch.all(params[1].grad == 1))
    # the batched report of a grouped weight-gradient flush (csrc/torch_ops.cpp notify_many): one call, every parameter counted once
    sync.zero_grad()
    sync._on_grads_ready([params[1], params[0]])
    ok = ok and sync.launch_log == [0, 1] and sync.collectives_launched == 4
    sync.finish()
    sync.remove()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def test_always_reduce_runs_the_collectives_at_world_size_1():
    """round 3: `always_reduce=True` is how ONE GPU drives the real N > 1 step through RCCL (bench.py --rccl-ws1,
    tests/test_gpu_grad_sync.py); here the same switch and the batched gradient-ready report over gloo on the CPU"""
    import torch.multiprocessing as mp
    mp.spawn(_worker_ws1, args=(1, _free_port(), None), nprocs=1, join=True)


def test_flat_parameters_give_the_same_adamw_step_as_per_tensor_parameters():
    """flat_parameters(): the registered parameters become views of one arena per dtype; AdamW over the flat leaf must move every
    parameter exactly as AdamW over the individual tensors does (the update is elementwise), including a channels_last conv factor."""
    from lycoris_amd.grad_sync import AdapterGradSync
    torch.manual_seed(3)
    shapes = [(8, 8), (160, 16), (3,), (12, 6, 3, 3)]
    ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    ref[3] = torch.nn.Parameter(ref[3].detach().contiguous(memory_format=torch.channels_last))
    mine = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in ref]
    sync = AdapterGradSync(mine)
    flats = sync.flat_parameters()
    assert len(flats) == 1 and flats[0].grad.data_ptr() == sync.arenas[torch.float32].data_ptr()
    assert all(p.data.untyped_storage().data_ptr() == flats[0].data.untyped_storage().data_ptr() for p in mine)
    assert mine[3].is_contiguous(memory_format=torch.channels_last) and all(torch.equal(a, b) for a, b in zip(mine, ref))
    o_ref = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.1)
    o_mine = torch.optim.AdamW(flats, lr=1e-2, weight_decay=0.1)
    for step in range(3):
        sync.zero_grad()
        o_ref.zero_grad()
        for plist in (ref, mine):
            sum(((p * (i + 1 + step)) ** 2).sum() for i, p in enumerate(plist)).backward()
        sync.finish()
        o_ref.step()
        o_mine.step()
        for a, b in zip(mine, ref):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), step
    sd = {f"p{i}": p for i, p in enumerate(mine)}
    assert all(v.shape == r.shape for v, r in zip(sd.values(), ref))


def test_the_last_bucket_to_complete_is_kept_small():
    """the collective of the input-side layers' bucket overlaps nothing (the backward pass is over when it starts): the trailing
    `tail_bucket_bytes` of the arena are a bucket of their own"""
    from lycoris_amd.grad_sync import AdapterGradSync
    params = [torch.nn.Parameter(torch.zeros(1000)) for _ in range(40)]  # 4 000 bytes each, registration = forward order
    sync = AdapterGradSync(params, bucket_bytes=64000, tail_bucket_bytes=9000)
    sizes = [b.flat.numel() * 4 for b in sync.buckets]
    assert sum(sizes) == 160000 and sizes[-1] == 8000 and len(sync.buckets[-1].params) == 2
    assert sync.buckets[-1].params == [params[1], params[0]]  # the first layers of the network: the last gradients of a backward pass
    assert all(sync._bucket_of[p] is b for b in sync.buckets for p in b.params) and len(sync._bucket_of) == 40
    assert [b.index for b in sync.buckets] == list(range(len(sync.buckets)))
    sum((p * (i + 1)).sum() for i, p in enumerate(params)).backward()
    assert sorted(sync.launch_log) == list(range(len(sync.buckets)))
    sync.finish()
    assert all(torch.all(p.grad == i + 1) for i, p in enumerate(params))
    nosplit = AdapterGradSync([torch.nn.Parameter(torch.zeros(1000)) for _ in range(40)], bucket_bytes=64000, tail_bucket_bytes=0)
    assert len(nosplit.buckets) == len(sync.buckets) - 1


def _worker_sharded(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lycoris_amd.grad_sync import AdapterGradSync, ShardedAdamW
        torch.manual_seed(0)  # identical replicas
        shapes = [(8, 8), (161, 16), (3,), (40, 5), ()]  # odd sizes: shards and leftover elements both exercised
        mine = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
        ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
        sync = AdapterGradSync(mine, bucket_bytes=4000, collective="reduce_scatter", tail_bucket_bytes=0)
        opt = ShardedAdamW(sync, lr=1e-2, weight_decay=0.1)
        o_ref = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.1)
        ok = True
        for step in range(3):
            opt.zero_grad()
            torch.manual_seed(100 * step + rank)
            noise = [torch.randn_like(p) for p in mine]
            sum((p * n).sum() + (p ** 2).sum() for p, n in zip(mine, noise)).backward()
            sync.finish()
            opt.step()
            # reference: the same loss on every rank's data, gradients averaged by hand, plain AdamW
            o_ref.zero_grad()
            acc = None
            for r in range(world):
                torch.manual_seed(100 * step + r)
                ns = [torch.randn_like(p) for p in ref]
                gs = torch.autograd.grad(sum((p * n).sum() + (p ** 2).sum() for p, n in zip(ref, ns)), ref)
                acc = list(gs) if acc is None else [a + g for a, g in zip(acc, gs)]
            for p, g in zip(ref, acc):
                p.grad = g / world
            o_ref.step()
            ok = ok and all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(mine, ref))
        out.put((rank, ok, len(sync.buckets)))
    finally:
        dist.destroy_process_group()


def test_sharded_adamw_equals_adamw_on_the_averaged_gradients():
    """ShardedAdamW: reduce-scatter during backward, every rank updates its shard of the flat parameter arena, the PARAMETERS are
    all-gathered -- the replicas must stay equal to a plain AdamW run on the hand-averaged gradients (2 ranks, 3 steps)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(2))
    assert [g[1] for g in got] == [True, True] and got[0][2] >= 2, got


def test_sharded_adamw_single_process_is_plain_adamw():
    from lycoris_amd.grad_sync import AdapterGradSync, ShardedAdamW
    torch.manual_seed(1)
    mine = [torch.nn.Parameter(torch.randn(s)) for s in [(8, 8), (33,)]]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    sync = AdapterGradSync(mine, collective="reduce_scatter")
    opt, o_ref = ShardedAdamW(sync, lr=1e-2), torch.optim.AdamW(ref, lr=1e-2)
    for step in range(2):
        opt.zero_grad(); o_ref.zero_grad()
        for plist in (mine, ref):
            sum(((p + step) ** 3).sum() for p in plist).backward()
        sync.finish(); opt.step(); o_ref.step()
        assert all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(mine, ref))


def test_sharded_adamw_state_is_rank_local_and_refuses_another_layout():
    """the moments belong to this rank's shards -- the state dict carries the layout it was cut for and a mismatching one
    is refused; 16-bit parameter arenas are refused (the moments would silently be 16-bit)"""
    from lycoris_amd.grad_sync import AdapterGradSync, ShardedAdamW
    torch.manual_seed(2)
    mine = [torch.nn.Parameter(torch.randn(s)) for s in [(8, 8), (33,)]]
    sync = AdapterGradSync(mine, collective="reduce_scatter")
    opt = ShardedAdamW(sync, lr=1e-2)
    opt.zero_grad()
    sum((p ** 2).sum() for p in mine).backward()
    sync.finish(); opt.step()
    st = opt.state_dict()
    assert st["layout"] == {"world": 1, "rank": 0, "shard_numels": [97]}
    other = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    sync2 = AdapterGradSync(other, collective="reduce_scatter")
    opt2 = ShardedAdamW(sync2, lr=1e-2)
    opt2.load_state_dict(st)                                   # same layout: accepted, moments restored
    m1 = next(iter(opt.inner.state.values()))["exp_avg"]
    m2 = next(iter(opt2.inner.state.values()))["exp_avg"]
    assert torch.equal(m1, m2)
    bad = dict(st, layout=dict(st["layout"], world=2))
    with pytest.raises(ValueError, match="layout"):
        opt2.load_state_dict(bad)
    half = [torch.nn.Parameter(torch.randn(16).bfloat16())]
    with pytest.raises(TypeError, match="fp32"):
        ShardedAdamW(AdapterGradSync(half, collective="reduce_scatter"))


def test_overlap_rule_is_payload_aware():
    """captured steps overlap their collectives when the estimated ring time exceeds the cost of the second queue"""
    from lycoris_amd.grad_sync import exchange_estimate_ms, overlap_pays
    assert exchange_estimate_ms(160 << 20, 1) == 0.0 and not overlap_pays(787 << 20, 1)
    assert abs(exchange_estimate_ms(250_000_000, 8) - 1.75) < 1e-9
    assert overlap_pays(787 << 20, 2) and overlap_pays(787 << 20, 8)          # LoHa: always
    assert not overlap_pays(25 << 20, 8) and not overlap_pays(153 << 20, 2)    # LoKr rank 16; LoKr full matrix on two GPUs
    assert overlap_pays(153 << 20, 8)                                          # ... on eight: marginal (1.1 ms against the 1.0 ms hop)
    assert not overlap_pays(153 << 20, 8, hop_ms=2.0) and overlap_pays(153 << 20, 2, busbw_gbs=100.0)


# ---- round 5: the ProcessGroup-free communicator path (host logic; the real thing runs in tests/test_gpu_grad_sync.py) -----------------
class _FakeComm:
    """records what AdapterGradSync asks of a communicator; `world` ranks, this is rank `rank`"""
    SUM, AVG, MAX = 0, 1, 2

    def __init__(self, world=2, rank=0, on_current_stream=False):
        self.world, self.rank, self.device, self.on_current_stream = world, rank, torch.device("cpu"), on_current_stream
        self.log = []

    def wait_current(self): self.log.append(("wait_current",))
    def wait_event(self, e): self.log.append(("wait_event", e))
    def mark(self): self.log.append(("mark",)); return len(self.log)
    def join(self): self.log.append(("join",))
    def all_reduce(self, t, op=0): self.log.append(("all_reduce", t.numel(), op)); t.mul_(0.5) if op == 1 else None
    def reduce_scatter(self, shard, full, op=0): self.log.append(("reduce_scatter", shard.numel(), full.numel(), op))
    def all_gather(self, full, shard): self.log.append(("all_gather", full.numel(), shard.numel()))


def test_communicator_path_orders_every_bucket_behind_its_gradients_and_joins_once():
    from lycoris_amd.grad_sync import AdapterGradSync
    params = [torch.nn.Parameter(torch.ones(1000)) for _ in range(6)]
    comm = _FakeComm()
    sync = AdapterGradSync(params, bucket_bytes=8000, comm=comm)   # 3 buckets of 2 parameters, no process group anywhere
    assert sync.world_size == 2 and sync.side_stream is None and len(sync.buckets) == 3
    sync.zero_grad()
    loss = sum((p * (i + 1)).sum() for i, p in enumerate(params))
    loss.backward()
    # the hooks fired during backward: every bucket = wait for the compute stream, then ONE in-place all-reduce (AVG)
    assert [e[0] for e in comm.log] == ["wait_current", "all_reduce"] * 3 and all(e[2] == comm.AVG for e in comm.log if e[0] == "all_reduce")
    sync.finish()
    assert comm.log[-1] == ("join",) and [e[0] for e in comm.log].count("join") == 1
    assert torch.allclose(params[0].grad, torch.full((1000,), 0.5))  # (the fake "averages" by halving)
    # segment callers: a mark of the compute stream, collectives ordered behind exactly that mark
    comm.log.clear()
    sync.zero_grad()
    m = sync.mark()
    sync.launch_buckets([0, 2], after=m)
    assert [e[0] for e in comm.log] == ["mark", "wait_event", "all_reduce", "wait_event", "all_reduce"] and comm.log[1][1] == m
    sync.finish()  # launches the bucket nobody asked for, joins
    assert [e[0] for e in comm.log][-3:] == ["wait_current", "all_reduce", "join"]


def test_communicator_path_reduce_scatter_shapes_and_unique_id_exchange(monkeypatch):
    import threading
    import types
    import lycoris_amd.grad_sync as gs
    from lycoris_amd import _native
    params = [torch.nn.Parameter(torch.ones(1001))]
    comm = _FakeComm(world=2, rank=1)
    sync = gs.AdapterGradSync(params, comm=comm, collective="reduce_scatter")
    sync.zero_grad()
    params[0].sum().backward()
    sync.finish()
    ops_ = [e for e in comm.log if e[0] in ("reduce_scatter", "all_gather", "all_reduce")]
    assert ops_ == [("reduce_scatter", 500, 1000, comm.AVG), ("all_gather", 1000, 500), ("all_reduce", 1, comm.AVG)]
    # the 128-byte id: rank 0 publishes it under the n-th key of the process, the others block on the same key
    made = []
    fake = types.SimpleNamespace(rccl_unique_id=lambda: b"U" * 128,
                                 RcclComm=lambda uid, rank, world, idx, hp, ext, cur: made.append((uid, rank, world, cur)))
    monkeypatch.setattr(_native, "load_torch_ops", lambda: fake)
    monkeypatch.setattr(torch.cuda, "init", lambda: None)

    class Store(dict):
        cv = threading.Condition()

        def set(self, k, v):
            with self.cv:
                self[k] = v
                self.cv.notify_all()

        def get(self, k):
            with self.cv:
                assert self.cv.wait_for(lambda: k in self, timeout=10)
                return self[k]

    st = Store()
    monkeypatch.setattr(gs.RcclCommunicator, "_created", 0)
    gs.RcclCommunicator(0, 2, "cuda:0", store=st, on_current_stream=True)   # rank 0 publishes
    monkeypatch.setattr(gs.RcclCommunicator, "_created", 0)                  # (another process: its own count)
    gs.RcclCommunicator(1, 2, "cuda:0", store=st)                            # rank 1 reads the same key
    assert sorted(m[1] for m in made) == [0, 1] and all(m[0] == b"U" * 128 for m in made) and made[0][3] is True
    assert list(st) == ["lycoris_amd/rccl_uid/1"]
    with pytest.raises(ValueError, match="store"):
        gs.RcclCommunicator(0, 2, "cuda:0")


_FROM_ENV_WORKER = '''
import json, os, sys, types
import torch
sys.path.insert(0, {root!r})
import lycoris_amd.grad_sync as gs
from lycoris_amd import _native
ext = _native.load_torch_ops()                       # the real extension: ncclGetUniqueId needs no GPU
made = []
fake = types.SimpleNamespace(rccl_unique_id=ext.rccl_unique_id,
                             RcclComm=lambda uid, rank, world, idx, hp, stream, cur: made.append((uid, rank, world, idx, cur)))
_native.load_torch_ops = lambda: fake
torch.cuda.init = lambda: None
a = gs.RcclCommunicator.from_env("cuda:%s" % os.environ["LOCAL_RANK"], on_current_stream=True)
b = gs.RcclCommunicator.from_env("cuda:%s" % os.environ["LOCAL_RANK"])   # a second communicator of the process: its own key, its own id
assert (a.rank, a.world) == (int(os.environ["RANK"]), 2) and not torch.distributed.is_initialized()
with open(os.path.join({out!r}, "rank%d.json" % a.rank), "w") as f:
    json.dump([[m[0].hex(), m[1], m[2], m[3], m[4]] for m in made], f)
'''


def test_from_env_under_the_launcher_two_ranks_agree_on_the_id_without_a_process_group(tmp_path):
    """`bench.py --gpus N` as the driver starts it (python -m torch.distributed.run ... on 127.0.0.1) with --backend rccl: the 128-byte
    id goes from rank 0 to the others through the LAUNCHER's store (torch.distributed.rendezvous("env://")), no process group is ever
    created.  Two real ranks, the real ncclGetUniqueId; only ncclCommInitRank (needs two GPUs) is replaced by a recorder."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_FROM_ENV_WORKER.format(root=root, out=str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in (0, 1))
    assert len(r0) == len(r1) == 2
    for (u0, k0, w0, i0, c0), (u1, k1, w1, i1, c1) in zip(r0, r1):
        assert u0 == u1 and len(bytes.fromhex(u0)) == 128 and (k0, k1, w0, w1, i0, i1) == (0, 1, 2, 2, 0, 1) and c0 == c1
    assert r0[0][0] != r0[1][0] and r0[0][4] is True and r0[1][4] is False   # two communicators, two ids
