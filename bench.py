#!/usr/bin/env python3
"""Driver benchmark: SDXL UNet adapter train step (bs=1/GPU), LoKr factor=8, bf16, on N MI355X.

One "step" = one pass of the adapter hot path over the 788 adapted layers of the SDXL UNet at 1024x1024
(synthetic activations of the real shapes, benchmarks/sdxl_shapes.py): adapter forward (delta) and backward
(dx + factor gradients, accumulated straight into the flat gradient arena) for every layer, the arena zero-fill,
the data-parallel mean all-reduce of the adapter gradients (N > 1, RCCL) and a fused AdamW update of the adapter
parameters.  The frozen UNet's own GEMMs / convolutions are not part of the hot path and are not timed.
The compute of a step is captured once in a hipGraph and replayed.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0 (contract in the task statement) including "roofline" and "cpu_baseline".
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from benchmarks.sdxl_shapes import algorithmic_bytes, layer_rows, sdxl_unet_layers  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md ("8.0 TB/s spec")
FACTOR = 8
ALGO_LABEL = {"lokr": "LoKr factor=8", "locon": "LoCon dim=16 conv_dim=8", "loha": "LoHa dim=32"}
# HBM bytes per launch of the dominant kernels from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
# runs, gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md; benchmarks/pmc_summary.py writes the numbers to
# profiles/): counters cannot be collected inside the timed run, so the committed measurement is quoted here.
# profiles/r01_pmc_bench_linear.txt (739 Linear layers, 2 eager passes): kron3 1.19x, dw2s 2.2x the algorithmic bytes
PMC_TRAFFIC = {"kron3": 11331968, "dw2s": 17305621,
               # LoCon (profiles/r01_pmc_bench_locon_linear.txt): bneck_kernel 10.12 MB, lowrank_tn_kernel 10.38 MB per launch
               "locon3": (2 * 10121569 + 10382384) // 3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--algo", default="lokr", choices=["lokr", "locon", "loha"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--layers", default="all", help="'all', 'linear' or 'conv' (development)")
    ap.add_argument("--profile-ops", action="store_true", help="development: torch.profiler table of one eager pass")
    ap.add_argument("--pmc-pass", type=int, default=0,
                    help="run N eager compute passes and exit (for rocprofv3 --pmc, which cannot sample inside graph replays)")
    return ap.parse_args()


class Layer:
    """One adapted layer of the workload: static input x, static upstream gradient g, fp32 adapter factors."""

    def __init__(self, spec, algo, dtype, dev, gen):
        from lycoris_amd import ops
        self.spec, self.algo = spec, algo
        M, I_eff, O = layer_rows(spec)
        if spec["kind"] == "linear":
            self.x = torch.randn(spec["M"], spec["I"], device=dev, dtype=dtype, generator=gen).requires_grad_(True)
            self.g = torch.randn(spec["M"], O, device=dev, dtype=dtype, generator=gen) / math.sqrt(O)
            cin, ksz = spec["I"], ()
        else:
            self.x = torch.randn(spec["B"], spec["C"], spec["H"], spec["W"], device=dev, dtype=dtype,
                                 generator=gen).requires_grad_(True)
            ho = (spec["H"] + 2 * spec["pad"] - spec["k"]) // spec["stride"] + 1
            self.g = torch.randn(spec["B"], O, ho, ho, device=dev, dtype=dtype, generator=gen) / math.sqrt(O)
            cin, ksz = spec["C"], (spec["k"], spec["k"])
        f32 = dict(device=dev, dtype=torch.float32, generator=gen)
        if algo == "lokr":  # factor=8, full-matrix w2 (lora_dim >= 10000): w1 [8,8], w2 [O/8, I/8(,k,k)]
            w2 = torch.randn(O // FACTOR, cin // FACTOR, *ksz, **f32) * 0.05
            if ksz:  # conv factor kept in channels_last memory: the implicit-GEMM kernels read / write it in place
                w2 = w2.contiguous(memory_format=torch.channels_last)
            self.params = [torch.randn(FACTOR, FACTOR, **f32) * 0.3, w2]
        elif algo == "locon":  # dim 16 / conv_dim 8
            r = 16 if not ksz or ksz == (1, 1) else 8
            down = torch.randn(r, cin, *ksz, **f32) * 0.05
            if ksz:  # channels_last lora_down: the implicit-GEMM kernels read / write it in place
                down = down.contiguous(memory_format=torch.channels_last)
            self.params = [down, torch.randn(O, r, *([1] * len(ksz)), **f32) * 0.05]
        else:  # loha dim 32
            r = 32
            kk = ksz[0] * ksz[1] if ksz else 1
            self.params = [torch.randn(O, r, **f32) * 0.1, torch.randn(r, cin * kk, **f32), torch.randn(O, r, **f32) * 0.1,
                           torch.randn(r, cin * kk, **f32)]
        self.params = [torch.nn.Parameter(p) for p in self.params]
        self.ops = ops

    def forward(self):
        ops, s, p = self.ops, self.spec, self.params
        if s["kind"] == "linear":
            if self.algo == "lokr":
                return ops.lokr_linear(self.x, p[0], p[1], 1.0)
            if self.algo == "locon":
                return ops.locon_linear(self.x, p[0], p[1], 1.0)
            return ops.loha_linear(self.x, p[0], p[1], p[2], p[3], 1.0)
        st, pd, dl = (s["stride"],) * 2, (s["pad"],) * 2, (1, 1)
        if self.algo == "lokr":
            return ops.lokr_conv2d(self.x, p[0], p[1], 1.0, st, pd, dl)
        if self.algo == "locon":
            return ops.locon_conv2d(self.x, p[0], p[1], 1.0, st, pd, dl)
        return ops.loha_conv2d(self.x, p[0], p[1], p[2], p[3], 1.0, (s["O"], s["C"], s["k"], s["k"]), st, pd, dl)


def build_workload(algo, dtype, dev, which):
    gen = torch.Generator(device=dev).manual_seed(1234)
    layers = []
    for spec in sdxl_unet_layers(1):
        if which == "linear" and spec["kind"] != "linear":
            continue
        if which == "conv" and spec["kind"] != "conv":
            continue
        # one Layer object (static tensors) per *distinct* shape; `count` instances share the activations but own
        # their parameters would cost 788 x activations; instead every instance has its own parameters + grads
        # (what DP all-reduces and AdamW updates) and shares the activation buffers of its shape.
        proto = Layer(spec, algo, dtype, dev, gen)
        layers.append((proto, spec["count"]))
    return layers


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]

    from lycoris_amd import ops
    from lycoris_amd.grad_sync import AdapterGradSync

    protos = build_workload(args.algo, dtype, dev, args.layers)
    # every one of the 788 layer instances owns its parameters (DP payload / optimizer state are full size)
    instances = []
    for proto, count in protos:
        for i in range(count):
            params = proto.params if i == 0 else [torch.nn.Parameter(p.detach().clone()) for p in proto.params]
            instances.append((proto, params))
    all_params = [p for _, ps in instances for p in ps]
    sync = AdapterGradSync(all_params, bucket_bytes=32 << 20)
    ops.fused_grad_accumulation(True)
    opt = torch.optim.AdamW(all_params, lr=1e-4, fused=True)
    n_layers = len(instances)

    def compute_pass():
        """adapter forward of every layer, then backward of every layer (reverse order), grads += into the arena"""
        outs = []
        for proto, params in instances:
            saved = proto.params
            proto.params = params
            outs.append((proto.forward(), proto))
            proto.params = saved
        # dx is computed and handed back; the factor gradients are accumulated into the arena by the kernels themselves.
        # (autograd.grad instead of .backward(): the layer instances of one shape share their activation buffers, and
        # .backward() would add an artificial "x.grad += dx" elementwise kernel per layer on top of the hot path.)
        for y, proto in reversed(outs):
            torch.autograd.grad(y, [proto.x], proto.g)

    if args.pmc_pass:
        for _ in range(args.pmc_pass):
            sync.zero_grad()
            compute_pass()
        torch.cuda.synchronize()
        return
    if args.profile_ops:
        from torch.profiler import ProfilerActivity, profile
        sync.zero_grad()
        compute_pass()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            compute_pass()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
        return

    # ---- capture one step's compute in a hipGraph -------------------------------------------------------------
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sync.zero_grad()
        compute_pass()  # eager warm-up (allocator, lazy init)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    sync.zero_grad()
    with torch.cuda.graph(graph):
        for arena in sync.arenas.values():
            arena.zero_()
        compute_pass()

    def step():
        graph.replay()
        if world > 1:  # mean all-reduce of the arena buckets on the side stream, joined before the optimizer
            sync.all_reduce_now()
            sync.finish()
        opt.step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps / elapsed  # whole-job: every rank processes its own batch (weak scaling)

    result = {
        "metric": "SDXL UNet adapter train steps/sec (bs=1/GPU), " + ALGO_LABEL[args.algo], "value": round(value, 3),
        "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": "SDXL UNet 1024x1024 bs=1/GPU, preset full: 788 adapted layers (739 Linear + 49 Conv2d), "
                        "adapter fwd+bwd + grad all-reduce + fused AdamW; frozen UNet ops not timed",
            "algo": args.algo, "factor": FACTOR if args.algo == "lokr" else None, "layers": n_layers,
            "adapter_params": sum(p.numel() for p in all_params), "dp_payload_mb": round(sync.payload_bytes / 2**20, 1),
            "parallelism": f"dp{world}", "graph": "hipGraph replay of the compute pass",
        },
    }
    if rank == 0 and world == 1 and not args.no_roofline:
        result["roofline"] = roofline(protos, args.algo, dtype, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.algo)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _timed_graph(issue, dev):
    """ms per replay of a hipGraph holding the launches `issue(stream_ptr)` makes; HIP events on the launch stream"""
    from lycoris_amd import _native as N
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sp = N.stream_ptr(dev)
        issue(sp)  # eager warm-up
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=st):
            issue(sp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps, best = 5, float("inf")
        gph.replay()
        for _ in range(3):
            e0.record(st)
            for _ in range(reps):
                gph.replay()
            e1.record(st)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
    return best


def roofline_locon(protos, dtype, dev):
    """LoCon: the three launches per adapted Linear layer (`bneck_kernel` forward, `bneck_kernel` backward-dx,
    `lowrank_tn_kernel` factor gradients), timed as two hipGraphs (all forward launches, all backward launches) with
    HIP events.  Algorithmic bytes: fwd x + y + t, bwd g + dx + dt, gradients g + x + t + dt, fp32 factors."""
    from lycoris_amd import _native as N
    esz = torch.empty((), dtype=dtype).element_size()
    code = N.dtype_code(dtype)
    calls, nbytes, n_layers = [], 0, 0
    for proto, count in protos:
        s = proto.spec
        if s["kind"] != "linear":
            continue
        M, I, O = layer_rows(s)
        down = proto.params[0].detach().contiguous()
        up = proto.params[1].detach().contiguous()
        r = down.shape[0]
        rows = proto.x.detach().reshape(-1, I)
        g = torch.randn(M, O, device=dev, dtype=dtype)
        y = torch.empty(M, O, device=dev, dtype=dtype)
        dx = torch.empty(M, I, device=dev, dtype=dtype)
        t = torch.empty(M, r, device=dev, dtype=torch.float32)
        dt = torch.empty(M, r, device=dev, dtype=torch.float32)
        dd, du = torch.zeros_like(down), torch.zeros_like(up)
        calls.append((count, rows, g, y, dx, down, up, t, dt, dd, du, (M, I, O, r)))
        fac = 4 * r * (I + O)
        nbytes += count * (esz * (M * I + M * O) + 4 * M * r + fac)            # forward
        nbytes += count * (esz * (M * O + M * I) + 4 * M * r + fac)            # backward dx
        nbytes += count * (esz * (M * O + M * I) + 8 * M * r + 2 * fac)        # factor gradients
        n_layers += count

    def fwd(sp):
        for count, rows, g, y, dx, down, up, t, dt, dd, du, (M, I, O, r) in calls:
            for _ in range(count):
                N.call("lyc_locon_linear_fwd", N.ptr(rows), N.ptr(down), N.ptr(up), N.ptr(t), N.ptr(y), M, I, O, r, 1.0, code, sp)

    def bwd(sp):
        for count, rows, g, y, dx, down, up, t, dt, dd, du, (M, I, O, r) in calls:
            for _ in range(count):
                N.call("lyc_locon_linear_bwd", N.ptr(g), N.ptr(rows), N.ptr(down), N.ptr(up), N.ptr(t), N.ptr(dt), N.ptr(dx),
                       N.ptr(dd), N.ptr(du), M, I, O, r, 1.0, code, sp)

    t_fwd, t_bwd = _timed_graph(fwd, dev), _timed_graph(bwd, dev)
    t_ms, n_launch = t_fwd + t_bwd, 3 * n_layers
    achieved = nbytes / (t_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "lyc::bneck_kernel (forward, backward-dx) + lyc::lowrank_tn_kernel (factor gradients): the "
                                      "three LoCon launches of the 739 Linear layers",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": PMC_TRAFFIC.get("locon3"), "launches_per_step": n_launch,
            "avg_launch_us": round(t_ms * 1e3 / n_launch, 2), "algorithmic_bytes_per_launch": int(nbytes / n_launch),
            "families_ms": {"forward": round(t_fwd, 3), "backward": round(t_bwd, 3)}}


# ------------------------------------------------------------------------------------------------------------------
def roofline(protos, algo, dtype, dev):
    """Roofline of the dominant kernel family of the step, measured live with HIP events on the launch stream.

    The LoKr step launches two kernel families per adapted Linear layer (Conv2d layers use the same kernels through
    the pixel-row gather and are left out of this leg: 49 of 788 layers):
      * `lyc::kron3_kernel`    -- forward, and backward dx (+ w1-gradient partials): HBM-bound (AI 20-200 flop/B);
      * `lyc::kron_dw2s_kernel` -- w2 gradient (split-K TN GEMM over the M*G rows).
    Three hipGraphs holding exactly the launches one step makes for the Linear layers (same shapes, same counts, the
    count instances rotate over distinct factor buffers) are replayed and timed: forward only, dW2 only, whole
    backward.  kron3 time = forward + (backward - dW2).  The family with the larger time is reported; `achieved` =
    algorithmic bytes of its launches / its time (activations moved once: fwd x + y, bwd-dx g + x + dx,
    dW2 g + x, plus the fp32 factors)."""
    from lycoris_amd import _native as N
    if algo == "locon":
        return roofline_locon(protos, dtype, dev)
    if algo != "lokr":
        return None
    esz = torch.empty((), dtype=dtype).element_size()
    code = N.dtype_code(dtype)
    calls, bytes_fwd, bytes_dx, bytes_dw2, n_layers = [], 0, 0, 0, 0
    for proto, count in protos:
        s = proto.spec
        if s["kind"] != "linear":
            continue
        M, I, O = layer_rows(s)
        w1 = proto.params[0].detach().contiguous()
        w2 = proto.params[1].detach().contiguous()
        a, b = w1.shape
        c, d = w2.shape
        rows = proto.x.detach().reshape(-1, I)
        g = torch.randn(M, O, device=dev, dtype=dtype)
        y = torch.empty(M, O, device=dev, dtype=dtype)
        dx = torch.empty(M, I, device=dev, dtype=dtype)
        dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
        ws = torch.empty(int(N.load().lyc_lokr_bwd_workspace_bytes(M, a, b, c, d, code)) + 16, dtype=torch.uint8, device=dev)
        calls.append((count, rows, g, y, dx, w1, w2, dw1, dw2, ws, (M, a, b, c, d)))
        fac = 4 * (a * b + c * d)
        bytes_fwd += count * (esz * (M * I + M * O) + fac)
        bytes_dx += count * (esz * (M * O + 2 * M * I) + fac)
        bytes_dw2 += count * (esz * (M * O + M * I) + fac)
        n_layers += count
    st = torch.cuda.Stream()

    def timed(issue):
        with torch.cuda.stream(st):
            sp = N.stream_ptr(dev)
            issue(sp)  # eager warm-up
            torch.cuda.synchronize()
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, stream=st):
                issue(sp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps, best = 5, float("inf")
            gph.replay()
            for _ in range(3):
                e0.record(st)
                for _ in range(reps):
                    gph.replay()
                e1.record(st)
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps)
        return best  # ms per pass over all Linear layers

    def fwd(sp):
        for count, rows, g, y, dx, w1, w2, dw1, dw2, ws, (M, a, b, c, d) in calls:
            for _ in range(count):
                N.call("lyc_lokr_linear_fwd", N.ptr(rows), N.ptr(w1), N.ptr(w2), N.ptr(y), M, a, b, c, d, 1.0, code, sp)

    def only_dw2(sp):
        for count, rows, g, y, dx, w1, w2, dw1, dw2, ws, (M, a, b, c, d) in calls:
            for _ in range(count):
                N.call("lyc_lokr_linear_bwd", N.ptr(g), N.ptr(rows), N.ptr(w1), N.ptr(w2), None, None, N.ptr(dw2), None,
                       M, a, b, c, d, 1.0, code, sp)

    def bwd(sp):
        for count, rows, g, y, dx, w1, w2, dw1, dw2, ws, (M, a, b, c, d) in calls:
            for _ in range(count):
                N.call("lyc_lokr_linear_bwd", N.ptr(g), N.ptr(rows), N.ptr(w1), N.ptr(w2), N.ptr(dx), N.ptr(dw1),
                       N.ptr(dw2), N.ptr(ws), M, a, b, c, d, 1.0, code, sp)

    t_fwd, t_dw2, t_bwd = timed(fwd), timed(only_dw2), timed(bwd)
    t_k3 = t_fwd + max(t_bwd - t_dw2, 0.0)
    fam = {
        "kron3": ("lyc::kron3_kernel (LoKr forward + backward-dx/dW1 launches of the 739 Linear layers)", t_k3,
                  bytes_fwd + bytes_dx, 2 * n_layers),
        "dw2s": ("lyc::kron_dw2s_kernel (LoKr dW2 launches of the 739 Linear layers)", t_dw2, bytes_dw2, n_layers),
    }
    key = "kron3" if t_k3 >= t_dw2 else "dw2s"
    name, t_ms, nbytes, n_launch = fam[key]
    achieved = nbytes / (t_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": PMC_TRAFFIC.get(key),
            "launches_per_step": n_launch, "avg_launch_us": round(t_ms * 1e3 / n_launch, 2),
            "algorithmic_bytes_per_launch": int(nbytes / n_launch),
            "families_ms": {"kron3_fwd": round(t_fwd, 3), "kron3_bwd": round(max(t_bwd - t_dw2, 0.0), 3),
                            "dw2s": round(t_dw2, 3)},
            "hot_path_gbs": round((bytes_fwd + bytes_dx) / ((t_fwd + t_bwd) * 1e-3) / 1e9, 1)}


def cpu_baseline(algo):
    """The reference's CPU path (rebuild: kron -> dense op -> autograd), restated in oracle/torch_cpu.py, timed on the
    host cores of this box on a bounded sample: one instance of every distinct Linear shape and of the 3x3/1x1 conv
    shapes with <= 4096 output pixels, fp32; the step time is extrapolated with the per-shape counts (the skipped
    large-conv shapes are charged at the cost of the largest measured conv per output pixel)."""
    from oracle import torch_cpu
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)

    def factors(spec):
        if spec["kind"] == "linear":
            cin, ksz, O = spec["I"], (), spec["O"]
        else:
            cin, ksz, O = spec["C"], (spec["k"], spec["k"]), spec["O"]
        if algo == "lokr":
            return [torch.randn(FACTOR, FACTOR) * 0.3, torch.randn(O // FACTOR, cin // FACTOR, *ksz) * 0.05]
        if algo == "locon":
            r = 16 if not ksz or ksz == (1, 1) else 8
            return [torch.randn(r, cin, *ksz) * 0.05, torch.randn(O, r, *([1] * len(ksz))) * 0.05]
        kk = ksz[0] * ksz[1] if ksz else 1
        return [torch.randn(O, 32) * 0.1, torch.randn(32, cin * kk), torch.randn(O, 32) * 0.1, torch.randn(32, cin * kk)]

    t_start = time.perf_counter()
    total, measured, skipped_px, px_cost = 0.0, 0, 0, 0.0
    for spec in sdxl_unet_layers(1):
        M, _, _ = layer_rows(spec)
        if spec["kind"] == "conv" and (M > 4096 or time.perf_counter() - t_start > 25.0):
            skipped_px += spec["count"] * M * spec["C"] * spec["O"] * spec["k"] ** 2
            continue
        t = torch_cpu.time_layer(algo, spec, factors, torch.float32, reps=1)
        total += t * spec["count"]
        measured += 1
        if spec["kind"] == "conv":
            px_cost = max(px_cost, t / (M * spec["C"] * spec["O"] * spec["k"] ** 2))
    total += skipped_px * px_cost
    return {"value": round(1.0 / total, 5), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/torch_cpu.py (reference rebuild path restated in torch CPU ops), fp32, {measured} distinct layer "
                      f"shapes timed once (best of 2) in {time.perf_counter() - t_start:.1f}s, extrapolated by shape counts "
                      "to the 788-layer step; conv shapes with > 4096 output pixels charged pro rata"}


if __name__ == "__main__":
    main()
