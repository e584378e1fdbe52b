"""Data-parallel synchronisation of the adapter gradients -- one process per GPU, RCCL over xGMI.

The reference has no code for this: under sd-scripts the network is wrapped in DistributedDataParallel, which copies
every gradient into bucket buffers and back.  Here the adapter gradients *live* in one flat arena per dtype:

  * every trainable adapter parameter's ``.grad`` is a view into the arena, so autograd accumulates in place and
    ``zero_grad()`` is one memset per arena instead of one per parameter;
  * the arena is cut into contiguous buckets in reverse registration order (backward produces the output-side
    layers' gradients first); when the last gradient of a bucket has been accumulated
    (``register_post_accumulate_grad_hook``) the bucket's arena slice is all-reduced (mean) **in place** with
    ``torch.distributed`` (backend "nccl" is RCCL on ROCm) on a dedicated side HIP stream, overlapping the rest of
    the frozen model's backward; no staging copies;
  * ``finish()`` joins the side stream (event wait, no host sync) before the optimizer step.
  * gradient accumulation: every backward but the last of an optimizer step runs inside ``no_sync()`` (same contract
    as DDP); a second un-guarded backward after a bucket has been reduced raises instead of letting replicas diverge.
  * the kernels' fused accumulation (``ops.fused_grad_accumulation``) bypasses the autograd hooks; ``attach_fused()``
    routes its per-parameter notification to the same bucket counters, so the collectives are launched from inside
    the backward pass there too.

  * round 4: `collective="reduce_scatter"` runs every bucket as an in-place reduce-scatter + all-gather pair; the last bucket to
    become complete (the input-side layers: its collective overlaps nothing) is kept small (`tail_bucket_bytes`);
    `bucket_boundaries()` / `launch_buckets()` serve callers that replay the backward pass in captured segments;
    `flat_parameters()` re-homes the parameters into a twin arena so that an optimizer steps ONE flat leaf per dtype, and
    `ShardedAdamW` shards the optimizer step and the second half of the exchange over the ranks (ZeRO-1 for the adapters).

The frozen base model is never touched: only adapter parameters are registered.  Do NOT also wrap the network in
DDP.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by one link, so buckets are
kept large (default 32 MiB: ~0.4 ms on a ring) and few; SDXL payloads are 25-790 MB.

  * round 5: `comm=RcclCommunicator(...)` takes the exchange off c10d altogether ("a separate ProcessGroup-free RCCL
    communicator object created once per process"): ncclCommInitRank once, the bucket collectives are plain work on the
    communicator's own HIP stream ordered by events (csrc/rccl_comm.cpp) -- no Work objects, no watchdog thread, capturable.

Works on CPU tensors with the gloo backend too (that is how tests/test_grad_sync.py covers the N > 1 path).
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass, field
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


# ---- where do the bucket collectives of a REPLAYED (captured) step go? ---------------------------------------------------------------
# Measured (profiles/r05_ws1_*): while a second stream waits on the stream that launches hipGraphs, every graph launch gets slower --
# about 1.0 ms on the captured 18 ms SDXL step, whatever the collectives cost themselves.  That is the price of overlapping; what it
# buys is the ring time of the exchange.  Never measured here (single-GPU boxes): RCCL's large-message all-reduce bus bandwidth on the
# 8-GPU xGMI mesh, taken as 250 GB/s (7 links x ~153 GB/s per GPU, ring collectives are per-link bound; the driver's SCALE run is the
# first measurement).  The rule, not a hard-wired default: overlap when the estimated ring time exceeds the hop.
XGMI_ALLREDUCE_BUSBW_GBS = 250.0
SIDE_STREAM_HOP_MS = 1.0


def exchange_estimate_ms(payload_bytes: int, world: int, busbw_gbs: float = XGMI_ALLREDUCE_BUSBW_GBS) -> float:
    """ring all-reduce (or reduce-scatter + all-gather) of `payload_bytes` over `world` ranks: 2 (N - 1) / N * S / busbw"""
    if world <= 1:
        return 0.0
    return 2.0 * (world - 1) / world * payload_bytes / (busbw_gbs * 1e9) * 1e3


def overlap_pays(payload_bytes: int, world: int, hop_ms: float = SIDE_STREAM_HOP_MS, busbw_gbs: float = XGMI_ALLREDUCE_BUSBW_GBS) -> bool:
    """captured steps: bucket collectives on the communicator's own stream between the backward segments (True) or on the compute
    stream behind the backward graph (False)?  SDXL at N = 8: LoKr full-matrix 153 MB -> 1.07 ms (overlap, marginal), LoKr rank 16
    25 MB -> 0.17 ms (inline), LoCon 185 MB -> 1.3 ms, LoHa 787 MB -> 5.5 ms (overlap); at N = 2 only LoHa overlaps."""
    return exchange_estimate_ms(payload_bytes, world, busbw_gbs) > hop_ms


class RcclCommunicator:
    """A ProcessGroup-free RCCL communicator (csrc/rccl_comm.cpp): one per process and GPU, created once.

        comm = RcclCommunicator.from_env(device)            # under torchrun / any env:// launcher: id through the launcher's store
        comm = RcclCommunicator(rank, world, device, store) # any object with set(key, bytes) / get(key) -> bytes
        comm = RcclCommunicator.from_process_group()        # id broadcast through an existing (e.g. gloo) group
        sync = AdapterGradSync(params, comm=comm)

    Collectives are enqueued on the communicator's own HIP stream (default priority: a high-priority stream made every kernel of the
    step 3.5x slower, profiles/r05_ws1_stream_and_event_ab.log; `high_priority=True` is the A/B switch) or, with
    `on_current_stream=True`, on whatever stream is current at the call (no second queue: what a caller that replays captured steps
    wants for small payloads, see `overlap_pays`).  `wait_current()` / `wait_event()` order them behind the producer of the data,
    `join()` makes the caller's stream wait for them.  Nothing here synchronises the host."""

    SUM, AVG, MAX = 0, 1, 2
    _created = 0

    def __init__(self, rank: int, world: int, device, store=None, unique_id: Optional[bytes] = None, key: str = "lycoris_amd/rccl_uid",
                 high_priority: bool = False, stream: Optional["torch.cuda.Stream"] = None, on_current_stream: bool = False):
        from . import _native
        ext = _native.load_torch_ops()
        device = torch.device(device)
        if device.type != "cuda":
            raise ValueError("RcclCommunicator needs a HIP device")
        index = device.index if device.index is not None else torch.cuda.current_device()
        if unique_id is None:
            if store is None:
                if world != 1:
                    raise ValueError("RcclCommunicator: world > 1 needs a store (or the unique id) to agree on the communicator")
                unique_id = ext.rccl_unique_id()
            else:
                # every rank creates its communicators in the same order: the n-th one of a process uses the n-th key
                RcclCommunicator._created += 1
                key = f"{key}/{RcclCommunicator._created}"
                if rank == 0:
                    unique_id = ext.rccl_unique_id()
                    store.set(key, unique_id)
                else:
                    unique_id = bytes(store.get(key))  # blocks until rank 0 has published it
        torch.cuda.init()
        self._stream = stream  # (kept alive: the communicator enqueues on it)
        # on_current_stream: every collective is enqueued on whatever stream is current at the call -- ordinary in-order work of the
        # compute stream, no events, no second HIP queue.  Measured on the captured SDXL step (profiles/r05_ws1_*): while a SECOND
        # stream waits on an event of the stream that launches hipGraphs, every graph launch gets slower (host 2.06 -> 2.75 ms per
        # 900-node graph; ~1 ms on the 18.4 ms step) -- about the ring time of the 153 MB exchange the side stream would hide.  Callers
        # that replay captured steps therefore default to this mode; eager training (long frozen backward, kernels launched one by
        # one: a hop costs 25 us) keeps the communicator's own stream and overlaps.
        self.on_current_stream = bool(on_current_stream)
        self._c = ext.RcclComm(unique_id, int(rank), int(world), int(index), bool(high_priority), 0 if stream is None else int(stream.cuda_stream),
                               self.on_current_stream)
        self.rank, self.world, self.device = int(rank), int(world), torch.device("cuda", index)

    @classmethod
    def from_env(cls, device, key: str = "lycoris_amd/rccl_uid", **kw):
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torchrun sets them; the id travels through the launcher's own store
        (torch.distributed.rendezvous: no process group is created)."""
        import os
        if int(os.environ.get("WORLD_SIZE", "1")) == 1 and "MASTER_PORT" not in os.environ:
            return cls(0, 1, device, **kw)
        store, rank, world = next(iter(dist.rendezvous("env://")))
        return cls(rank, world, device, store=store, key=key, **kw)

    @classmethod
    def from_process_group(cls, group=None, device=None):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        from . import _native
        box = [_native.load_torch_ops().rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, device, unique_id=box[0])

    # ---- ordering ------------------------------------------------------------------------------------------------------------
    def wait_current(self):
        self._c.wait_current()

    def wait_event(self, event):
        """`event`: a mark (int, from mark()) or a torch.cuda.Event recorded on a stream of this device"""
        if isinstance(event, int):
            self._c.wait_mark(event)
        else:
            self._c.wait_event(event.cuda_event)

    def mark(self) -> int:
        """remember this point of the current stream (a device-scope event of the communicator's ring of 64); wait_event(mark) later
        orders the communicator's stream behind exactly this point"""
        return self._c.mark()

    def join(self):
        self._c.join()

    def synchronize(self):
        self._c.synchronize()

    # ---- collectives (in place, on the communicator's stream) ------------------------------------------------------------------
    def all_reduce(self, t, op=0):
        self._c.all_reduce(t, op)

    def reduce_scatter(self, shard, full, op=0):
        self._c.reduce_scatter(shard, full, op)

    def all_gather(self, full, shard):
        self._c.all_gather(full, shard)

    def broadcast(self, t, root=0):
        self._c.broadcast(t, root)

    def barrier(self):
        """host-blocking: every rank's compute stream has reached this point"""
        if getattr(self, "_token", None) is None:
            self._token = torch.zeros(1, device=self.device)
        self.wait_current()
        self._c.all_reduce(self._token, 0)
        self._c.synchronize()

    def max_over_ranks(self, value: float) -> float:
        t = torch.tensor([value], device=self.device, dtype=torch.float64)
        self.wait_current()
        self._c.all_reduce(t, self.MAX)
        self._c.synchronize()
        return float(t)

    def count(self) -> int:
        """the number of ranks RCCL itself reports (ncclCommCount) -- not what the launcher's environment said"""
        return int(self._c.count())

    def destroy(self):
        self._c.destroy()


@dataclass(eq=False)
class _Bucket:
    flat: torch.Tensor            # slice of the arena
    n_params: int
    pending: int = 0
    work: Optional[object] = None
    launched: bool = False
    index: int = 0
    params: List[torch.nn.Parameter] = field(default_factory=list)


class AdapterGradSync:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20,
                 process_group=None, average: bool = True, always_reduce: bool = False, collective: str = "all_reduce",
                 tail_bucket_bytes: int = 0, comm: Optional[RcclCommunicator] = None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("AdapterGradSync: no trainable parameters")
        if collective not in ("all_reduce", "reduce_scatter"):
            raise ValueError("AdapterGradSync: collective must be 'all_reduce' or 'reduce_scatter'")
        # "reduce_scatter": every bucket as an in-place reduce-scatter + all-gather pair (each rank reduces 1 / N of the
        # bucket; same bytes on the wire as a ring all-reduce, but the two halves are separate collectives that RCCL schedules over
        # all xGMI links and a sharded optimizer could run between them).  Same result as "all_reduce" up to summation order.
        self.collective = collective
        self.group = process_group
        self.average = average
        # `comm`: the ProcessGroup-free communicator (round 5) -- the collectives are then plain stream work of `comm`, torch.distributed
        # is not touched on the data path (and need not be initialised at all)
        self.comm = comm
        if comm is not None:
            self.world_size = comm.world
        else:
            self.world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # `always_reduce`: issue the collectives even at world_size 1 (a process group or `comm` must exist) -- lets ONE GPU drive
        # the RCCL path end to end (benchmarks/rccl_ws1_check.py); a no-op all-reduce is otherwise skipped
        self._reduce = self.world_size > 1 or (always_reduce and (comm is not None or (dist.is_available() and dist.is_initialized())))
        dev = self.params[0].device
        if any(p.device != dev for p in self.params):
            raise ValueError("AdapterGradSync: all adapter parameters must live on one device")
        self.device = dev
        if comm is not None and comm.device != dev:
            raise ValueError(f"AdapterGradSync: the communicator is bound to {comm.device}, the parameters live on {dev}")
        self.side_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" and comm is None else None  # (`comm` owns its stream)
        self.arenas = {}
        self.buckets: List[_Bucket] = []
        self._bucket_of = {}
        self._handles = []
        self._sync_enabled = True
        self._fused = False
        self.collectives_launched = 0  # all-reduces issued since construction (diagnostics)
        self.launch_log: List[int] = []  # bucket indices in launch order of the current step (tests / diagnostics)
        # reverse registration order: the last layers' gradients are ready first during backward
        by_dtype = {}
        for p in reversed(self.params):
            by_dtype.setdefault(p.dtype, []).append(p)
        for dtype, plist in by_dtype.items():
            total = sum(p.numel() for p in plist)
            arena = torch.zeros(total, dtype=dtype, device=dev)
            self.arenas[dtype] = arena
            esz = arena.element_size()
            off = 0
            start, members = 0, []
            for p in plist:
                n = p.numel()
                # same memory layout as the parameter (e.g. channels_last conv factors): kernels that write
                # gradients in the parameter's own layout can then accumulate straight into the arena
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                p.grad = arena[off:off + n].as_strided(p.shape, p.stride()) if dense else arena[off:off + n].view_as(p)
                members.append(p)
                off += n
                if (off - start) * esz >= bucket_bytes:
                    self._close_bucket(arena, start, off, members)
                    start, members = off, []
            if members:
                self._close_bucket(arena, start, off, members)
            # The collective of the LAST bucket to become complete (the input-side layers) cannot overlap anything: the backward
            # pass is over when it starts.  Its duration is what a step pays for the whole exchange, so it CAN be kept small: the last
            # `tail_bucket_bytes` of the arena then form a bucket of their own.  Off by default: at world_size 1 it bought
            # nothing (21.58 ms with a 2 MiB tail against 21.45 ms without, profiles/r04_final_ws1.log) and costs one more collective
            # per step; whether it pays on a ring over xGMI needs a multi-GPU measurement.
            self._split_tail(arena, esz, tail_bucket_bytes)
        for p in self.params:
            self._handles.append(p.register_post_accumulate_grad_hook(self._on_autograd_hook))
        self._reset_pending()

    # ---- construction helpers --------------------------------------------------------------------------------
    # ---- flat parameters: the optimizer step as ONE elementwise pass ------------------------------------------------------------
    def flat_parameters(self):
        """Re-home every registered parameter into a flat arena per dtype (same order as the gradient arena: `p.data` becomes a view,
        values, shapes, strides and the Parameter objects themselves stay) and return one flat leaf per dtype whose `.grad` is the
        gradient arena.  An optimizer built on the returned list updates all adapter parameters in one multi-tensor chunk stream

            opt = torch.optim.AdamW(sync.flat_parameters(), lr=..., fused=True)

        instead of walking 1 576 small tensors (SDXL LoKr: 1.40 -> ~0.3 ms per step, bench.py).  All parameters of one arena share
        the optimizer hyper-parameters; layers that need their own (LoRA+ ratios, text-encoder learning rates) get their own
        AdapterGradSync, or a regular per-tensor optimizer.  state_dict() of the modules is unaffected (views serialise as
        tensors); the optimizer's own state is per flat tensor.  The update does not move the modules' version counters: the LoKr
        plane cache follows step boundaries (optimizer-step hook), not versions."""
        if getattr(self, "_flat", None) is not None:
            return self._flat
        flats = []
        for dtype, garena in self.arenas.items():
            parena = torch.empty_like(garena)
            off = 0
            with torch.no_grad():
                for p in reversed(self.params):
                    if p.dtype != dtype:
                        continue
                    n = p.numel()
                    dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                    view = parena[off:off + n].as_strided(p.shape, p.stride()) if dense else parena[off:off + n].view_as(p)
                    view.copy_(p)
                    p.data = view
                    off += n
            flat = torch.nn.Parameter(parena, requires_grad=True)
            flat.grad = garena
            flats.append(flat)
        self._flat = flats
        return flats

    def _split_tail(self, arena, esz, tail_bytes):
        if tail_bytes <= 0 or not self.buckets:
            return
        last = self.buckets[-1]
        if last.flat.untyped_storage().data_ptr() != arena.untyped_storage().data_ptr() or last.n_params < 2:
            return
        if last.flat.numel() * esz <= 2 * tail_bytes:
            return
        # parameters of the bucket in arena order; the tail = the trailing parameters whose sizes add up to <= tail_bytes (at least one)
        acc, cut = 0, len(last.params)
        for i in range(len(last.params) - 1, 0, -1):
            n = last.params[i].numel() * esz
            if acc + n > tail_bytes and cut < len(last.params):
                break
            acc += n
            cut = i
        if cut >= len(last.params) or cut == 0:
            return
        head, tail = last.params[:cut], last.params[cut:]
        start = last.flat.storage_offset()
        head_elems = sum(p.numel() for p in head)
        end = start + last.flat.numel()
        self.buckets.pop()
        for p in last.params:
            self._bucket_of.pop(p, None)
        self._close_bucket(arena, start, start + head_elems, head)
        self._close_bucket(arena, start + head_elems, end, tail)

    def _close_bucket(self, arena, start, end, members):
        b = _Bucket(flat=arena[start:end], n_params=len(members), params=list(members), index=len(self.buckets))
        for p in members:
            self._bucket_of[p] = b
        self.buckets.append(b)

    def _reset_pending(self):
        for b in self.buckets:
            b.pending = b.n_params
            b.work = None
            b.launched = False
        self.launch_log = []

    # ---- per-step API ------------------------------------------------------------------------------------------
    @property
    def payload_bytes(self) -> int:
        return sum(a.numel() * a.element_size() for a in self.arenas.values())

    def zero_grad(self):
        """One memset per arena.  Use this instead of optimizer.zero_grad(set_to_none=True), which would detach the
        gradient views from the arena."""
        self._forget_pass()
        for arena in self.arenas.values():
            arena.zero_()
        for p in self.params:  # re-attach if somebody replaced / dropped a .grad
            b = self._bucket_of[p]
            if p.grad is None or p.grad.untyped_storage().data_ptr() != b.flat.untyped_storage().data_ptr():
                self._reattach()
                break
        self._reset_pending()

    def _forget_pass(self):
        """Step boundary of the fused path: layers parked by a backward pass that raised are dropped (they would otherwise be
        added to the next step's gradients) and stale pending-accumulation counts are cleared."""
        if self.device.type == "cuda" and self._fused:
            from . import ops
            ops.discard_deferred()
            ops.reset_use_counts()

    def _reattach(self):
        for dtype, arena in self.arenas.items():
            off = 0
            for p in reversed(self.params):
                if p.dtype != dtype:
                    continue
                n = p.numel()
                # same memory layout as the parameter (e.g. channels_last conv factors): kernels that write
                # gradients in the parameter's own layout can then accumulate straight into the arena
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                p.grad = arena[off:off + n].as_strided(p.shape, p.stride()) if dense else arena[off:off + n].view_as(p)
                off += n

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context only accumulate into the arena; the backward
        that runs outside it (the last micro-batch) counts the gradients and launches the collectives."""
        prev, self._sync_enabled = self._sync_enabled, False
        try:
            yield
        finally:
            self._sync_enabled = prev

    def attach_fused(self, enabled: bool = True):
        """Let the kernels accumulate factor gradients straight into the arena (``ops.fused_grad_accumulation``) and
        report each finished parameter to the bucket counters, exactly like the autograd hook would: ONE report per parameter
        and backward pass, after its last accumulation -- a parameter shared by several layer calls is counted in the forward
        pass (csrc/torch_ops.cpp `expect()`) and reported when the last of its backward nodes has run."""
        from . import ops
        ops.fused_grad_accumulation(enabled, callback=self._on_grad_ready if enabled else None,
                                    batch_callback=self._on_grads_ready if enabled else None)
        self._fused = bool(enabled)

    def _on_autograd_hook(self, p):
        """post-accumulate-grad hook.  autograd runs a leaf's AccumulateGrad node -- and this hook -- even when the backward node
        returned NO gradient for it, which is what the kernels do when they accumulate into `.grad` themselves (torch >= 2.x;
        found by tests/test_gpu_parity_round4.py: every fused parameter of a loss.backward() was counted twice and its bucket
        all-reduced before the last gradient had arrived).  Parameters the kernels report are counted by THAT report only."""
        if self._fused and self.device.type == "cuda":
            from . import ops
            if ops.fused_reports(p):
                return
        self._on_grad_ready(p)

    def _on_grads_ready(self, params):
        """the reports of one grouped weight-gradient flush (csrc/torch_ops.cpp notify_many): one call instead of one per parameter"""
        if not self._sync_enabled:
            return
        for p in params:
            self._on_grad_ready(p)

    def _on_grad_ready(self, p):
        if not self._sync_enabled:
            return
        b = self._bucket_of.get(p)
        if b is None:
            return
        if b.launched:
            raise RuntimeError(
                "AdapterGradSync: a gradient arrived for a bucket that was already all-reduced in this step. "
                "With gradient accumulation run every backward but the last inside `with sync.no_sync():`, and call "
                "finish() (or zero_grad()) once per optimizer step.")
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket, after=None):
        b.launched = True
        self.launch_log.append(b.index)
        if not self._reduce:
            return
        self.collectives_launched += 1
        if self.comm is not None:
            # stream work only: the communicator's stream waits for the bucket's gradients, then runs the collective(s) in place
            if after is not None:
                self.comm.wait_event(after)
            else:
                self.comm.wait_current()
            self._comm_reduce(b.flat)
            self._comm_pending = True
        elif self.side_stream is not None:
            # the bucket's gradients were produced on the compute stream: order the collective after them (`after`: an event
            # recorded when they were complete -- work enqueued on the compute stream since then is NOT waited for)
            if after is not None:
                self.side_stream.wait_event(after)
            else:
                self.side_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.side_stream):
                b.work = self._all_reduce(b.flat)
        else:
            b.work = self._all_reduce(b.flat)

    def _all_reduce(self, flat):
        backend = dist.get_backend(self.group)
        if self.collective == "reduce_scatter":
            return self._reduce_scatter_all_gather(flat, backend)
        if self.average and backend == "nccl":
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.average:
            work.wait()
            flat.div_(self.world_size)
            return None
        return work

    def _comm_reduce(self, flat):
        """the bucket through the ProcessGroup-free communicator: all-reduce, or reduce-scatter + all-gather (`collective`)"""
        c = self.comm
        op = c.AVG if self.average else c.SUM
        if self.collective != "reduce_scatter":
            c.all_reduce(flat, op)
            return
        world, rank, n = c.world, c.rank, flat.numel()
        chunk = n // world
        if chunk > 0:
            main = flat[:chunk * world]
            shard = main[rank * chunk:(rank + 1) * chunk]
            c.reduce_scatter(shard, main, op)
            if not getattr(self, "_shard_only", False):
                c.all_gather(main, shard)
        if chunk * world < n:
            c.all_reduce(flat[chunk * world:], op)

    def _reduce_scatter_all_gather(self, flat, backend):
        """in place: rank r reduces elements [r * chunk, (r + 1) * chunk) of the bucket (its shard is a view of the bucket at that
        offset: the in-place form of both collectives), then every rank gathers all shards; the < world_size elements that do not
        divide go through a small all-reduce"""
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        n = flat.numel()
        chunk = n // world
        nccl = backend == "nccl"  # collectives of one communicator run in issue order on its stream: no waits between them
        avg_native = self.average and nccl
        op = dist.ReduceOp.AVG if avg_native else dist.ReduceOp.SUM
        work = None
        if chunk > 0:
            main = flat[:chunk * world]
            shard = main[rank * chunk:(rank + 1) * chunk]
            w = dist.reduce_scatter_tensor(shard, main, op=op, group=self.group, async_op=True)
            work = w
            if not nccl:
                w.wait()
                if self.average:
                    shard.div_(world)
            if not getattr(self, "_shard_only", False):  # (ShardedAdamW gathers the updated PARAMETERS instead of the gradients)
                work = dist.all_gather_into_tensor(main, shard, group=self.group, async_op=True)
                if not nccl:
                    work.wait()
        if chunk * world < n:
            tail = flat[chunk * world:]
            work = dist.all_reduce(tail, op=op, group=self.group, async_op=True)  # after the all-gather in the communicator's order
            if not nccl:
                work.wait()
                if self.average:
                    tail.div_(world)
        return work if nccl else None

    def bucket_boundaries(self, order):
        """For callers that replay the backward pass in captured segments (no hooks fire inside a hipGraph): `order[p]` = position
        of parameter p's layer in FORWARD order.  Returns, per bucket in launch order, the smallest position among its parameters
        -- the bucket is complete once backward has run down to that layer.  Cutting the segments at exactly these positions gives
        one segment per bucket (SDXL LoKr: 5) instead of an arbitrary number of equal ones."""
        return [min(order[p] for p in b.params) for b in self.buckets]

    def mark(self):
        """a point of the compute stream `launch_buckets(..., after=mark)` can order collectives behind: the communicator's own
        device-scope event when there is one (a default torch.cuda.Event does a system-scope release at record time: +0.3 ms of GPU
        time per event between two backward segments), else a recorded torch.cuda.Event"""
        if self.comm is not None:
            return self.comm.mark()
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def launch_buckets(self, indices, after=None):
        """launch the collectives of the given buckets (precomputed per segment by the caller: no scan, no host synchronisation --
        only the stream wait / collective enqueue of `_launch`).  `after`: an event recorded on the compute stream right behind the
        segment that completed these buckets; the caller can then submit the NEXT segment's graph first and enqueue the collectives
        while the GPU already runs it (the enqueue costs tens of microseconds of host time per collective, which otherwise sit
        between two graph launches with the GPU idle)."""
        for i in indices:
            b = self.buckets[i]
            if not b.launched:
                b.pending = 0
                self._launch(b, after)

    def all_reduce_now(self):
        """Launch the collective of every bucket that has not been launched in this step -- for callers whose backward
        does not fire the autograd hooks (fused accumulation straight into the arena, or a replayed hipGraph)."""
        for b in self.buckets:
            if not b.launched:
                b.pending = 0
                self._launch(b)

    def launch_ready(self, upto_param=None):
        """Graph-replay callers (no hooks fire inside a hipGraph): launch the collective of every not-yet-launched
        bucket whose parameters all come before ``upto_param`` in backward order (None = all of them)."""
        for b in self.buckets:
            if b.launched:
                continue
            if upto_param is not None and any(q is upto_param for q in b.params):
                break
            b.pending = 0
            self._launch(b)

    def finish(self):
        """Call after backward, before optimizer.step(): flushes buckets whose hooks did not all fire (unused
        parameters) and makes the compute stream wait for the collectives."""
        for b in self.buckets:
            if not b.launched:  # not every hook fired (unused parameters, no_sync micro-batches, graph replays)
                b.pending = 0
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()  # on nccl this is a stream-level wait, not a host sync
                b.work = None
        if self.comm is not None:
            if getattr(self, "_comm_pending", False):
                self.comm.join()  # the compute stream waits for the collectives (an event; no host synchronisation)
                self._comm_pending = False
        elif self.side_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
        self._reset_pending()
        if self.device.type == "cuda" and self._fused:
            from . import ops
            ops.reset_use_counts()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles.clear()


class ShardedAdamW:
    """AdamW with the optimizer work and the second half of the exchange sharded over the data-parallel ranks ("prefer
    reduce-scatter + all-gather shapes"; ZeRO-1 for the adapters).  Needs `AdapterGradSync(..., collective="reduce_scatter")` and
    its flat parameter arena:

        sync = AdapterGradSync(params, collective="reduce_scatter"); opt = ShardedAdamW(sync, lr=1e-4)
        loss.backward(); sync.finish(); opt.step()

    During backward every bucket is only REDUCE-SCATTERED (rank r ends up with the mean gradient of elements
    [r * chunk, (r + 1) * chunk) of the bucket); `step()` updates exactly those elements of the flat parameter arena -- moments
    kept for the own shard only: 1 / N of the optimizer state and of its HBM traffic -- and ALL-GATHERS the updated parameters.  Same
    bytes on the wire as the all-reduce, but the second half moves parameters after the update and can overlap the next forward's
    first layers.  The < world_size leftover elements of a bucket are all-reduced and updated redundantly on every rank.

    The update itself is `torch.optim.AdamW` (fused multi-tensor kernel on the GPU: ONE launch per step) over views of the rank's
    shards -- element for element the update of torch.optim.AdamW on the averaged gradients (tests/test_grad_sync.py, 2 ranks).
    One hyper-parameter set per arena, as with flat_parameters().  Because the parameters change through views of the arena (no
    version counter of a module parameter moves), `step()` marks the LoKr operand-plane cache dirty itself.

    Moments live in the parameter dtype (torch.optim.AdamW's rule); the adapter parameters of this path are fp32, 16-bit
    parameter arenas are refused rather than silently given 16-bit moments.  `state_dict()` is RANK-LOCAL: it holds the moments of
    this rank's shards only, together with the layout they belong to (world, rank, shard sizes), and `load_state_dict()` refuses a
    state saved under another world size / rank / bucket layout instead of mis-assigning moments.

    Status: exercised under 2-rank gloo on the CPU and through RCCL at world_size 1 on one GPU (tests/test_gpu_grad_sync.py); never
    measured on several GPUs."""

    def __init__(self, sync: AdapterGradSync, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if sync.collective != "reduce_scatter":
            raise ValueError("ShardedAdamW needs AdapterGradSync(collective='reduce_scatter')")
        self.sync = sync
        self.flats = sync.flat_parameters()
        if any(f.dtype != torch.float32 for f in self.flats):
            raise TypeError("ShardedAdamW keeps its moments in the parameter dtype: fp32 adapter parameters only")
        sync._shard_only = sync._reduce  # with one rank (and no forced collectives) the plain path below updates everything
        self.world = sync.world_size if sync._reduce else 1
        if sync.comm is not None:
            self.rank = sync.comm.rank if sync._reduce else 0
        else:
            self.rank = dist.get_rank(sync.group) if sync._reduce else 0
        self.gathers = []  # per bucket: (all elements that divide over the ranks, this rank's shard of them)
        shards = []
        for b in sync.buckets:
            flat = next(f for f in self.flats if f.dtype == b.flat.dtype)
            lo0 = b.flat.storage_offset()
            n = b.flat.numel()
            pslice = flat.data[lo0:lo0 + n]
            chunk = n // self.world
            own = (self.rank * chunk, (self.rank + 1) * chunk) if self.world > 1 else (0, n)
            tail = (chunk * self.world, n) if self.world > 1 else (n, n)
            for lo, hi in (own, tail):
                if hi > lo:
                    q = torch.nn.Parameter(pslice[lo:hi], requires_grad=True)  # a view: the update lands in the arena
                    q.grad = b.flat[lo:hi]
                    shards.append(q)
            if self.world > 1 and chunk > 0:
                main = pslice[:chunk * self.world]
                self.gathers.append((main, main[own[0]:own[1]]))
        on_gpu = sync.device.type == "cuda"
        self.inner = torch.optim.AdamW(shards, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **({"fused": True} if on_gpu else {}))
        self.param_groups = self.inner.param_groups
        self._layout = {"world": self.world, "rank": self.rank, "shard_numels": [int(q.numel()) for q in shards]}

    def state_dict(self):
        """rank-local: the moments of THIS rank's shards + the layout they were cut for"""
        return {"inner": self.inner.state_dict(), "layout": dict(self._layout)}

    def load_state_dict(self, state):
        if state.get("layout") != self._layout:
            raise ValueError(f"ShardedAdamW: state saved for layout {state.get('layout')}, this optimizer has {self._layout} "
                             "(world size, rank and bucket_bytes must match: the moments belong to this rank's shards)")
        self.inner.load_state_dict(state["inner"])

    @torch.no_grad()
    def step(self):
        self.inner.step()
        sync = self.sync
        if self.gathers:
            if sync.comm is not None:
                sync.comm.wait_current()  # the update ran on the current stream
                for main, shard in self.gathers:
                    sync.comm.all_gather(main, shard)
                sync.comm.join()
            else:
                works = [dist.all_gather_into_tensor(main, shard, group=sync.group, async_op=True) for main, shard in self.gathers]
                for w in works:
                    w.wait()  # nccl: a stream-level wait
        if sync.device.type == "cuda":
            from . import ops
            ops.mark_planes_dirty()  # parameters changed through views of the arena: no version counter of a module moved

    def zero_grad(self, set_to_none: bool = False):
        self.sync.zero_grad()

