"""Per-shape kernel time of one adapter algorithm on the SDXL layer list (development tool).

For every distinct SDXL layer shape: the forward launches and the backward launches of the native op, each replayed from a
hipGraph (kernel time only, no Python), with the algorithmic bytes (activations in + out, 16-bit) and the GB/s they imply.

    python benchmarks/shape_times.py --algo locon|lokr|loha [--layers linear|conv|all]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from benchmarks.sdxl_shapes import sdxl_unet_layers
from lycoris_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--algo", default="locon", choices=["locon", "lokr", "loha"])
ap.add_argument("--layers", default="all", choices=["all", "linear", "conv"])
ap.add_argument("--rank", type=int, default=16)
ap.add_argument("--conv-rank", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.bfloat16
ops.fused_grad_accumulation(False)


def graphed(fn, reps=8, iters=10):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for _ in range(reps):
            fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gph.replay()
    e0.record()
    for _ in range(iters):
        gph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / reps * 1e3


def factors(spec):
    conv = spec["kind"] == "conv"
    I, O = (spec["C"], spec["O"]) if conv else (spec["I"], spec["O"])
    k = spec["k"] if conv else 1
    tail = (k, k) if conv else ()
    if args.algo == "locon":
        r = args.conv_rank if conv and k > 1 else args.rank
        return [torch.randn(r, I, *tail, device=dev) * 0.05, torch.randn(O, r, *((1, 1) if conv else ()), device=dev) * 0.05]
    if args.algo == "lokr":
        w2 = torch.randn(O // 8, I // 8, *tail, device=dev) * 0.05
        if conv:
            w2 = w2.contiguous(memory_format=torch.channels_last)
        return [torch.randn(8, 8, device=dev) * 0.3, w2]
    r = 32
    return [torch.randn(O, r, device=dev) * 0.1, torch.randn(r, I * k * k, device=dev) * 0.1,
            torch.randn(O, r, device=dev) * 0.1, torch.randn(r, I * k * k, device=dev) * 0.1]


def call(spec, x, fs):
    if spec["kind"] == "linear":
        if args.algo == "locon":
            return ops.locon_linear(x, fs[0], fs[1], 1.0)
        if args.algo == "lokr":
            return ops.lokr_linear(x, fs[0], fs[1], 1.0)
        return ops.loha_linear(x, *fs, 1.0)
    st, pd, dl = (spec["stride"],) * 2, (spec["pad"],) * 2, (1, 1)
    if args.algo == "locon":
        return ops.locon_conv2d(x, fs[0], fs[1], 1.0, st, pd, dl)
    if args.algo == "lokr":
        return ops.lokr_conv2d(x, fs[0], fs[1], 1.0, st, pd, dl)
    return ops.loha_conv2d(x, *fs, 1.0, (spec["O"], spec["C"], spec["k"], spec["k"]), st, pd, dl)


tot_f = tot_b = tot_bytes = 0.0
print(f"{'shape':46s} {'n':>4s} | {'fwd us':>8s} {'GB/s':>6s} | {'bwd us':>8s} {'GB/s':>6s}")
for spec in sdxl_unet_layers(1):
    if args.layers != "all" and spec["kind"] != args.layers:
        continue
    if spec["kind"] == "linear":
        x = torch.randn(spec["M"], spec["I"], device=dev, dtype=dt, requires_grad=True)
        n_in, n_out = spec["M"] * spec["I"], spec["M"] * spec["O"]
    else:
        x = torch.randn(spec["B"], spec["C"], spec["H"], spec["W"], device=dev, dtype=dt)
        x = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ho = (spec["H"] + 2 * spec["pad"] - spec["k"]) // spec["stride"] + 1
        n_in, n_out = x.numel(), spec["B"] * spec["O"] * ho * ho
    fs = [f.requires_grad_(True) for f in factors(spec)]
    g = torch.randn_like(call(spec, x, fs))
    tf = graphed(lambda: call(spec, x, fs))
    # backward = (forward + backward) - forward: autograd replays the backward on the stream the forward ran on, so both
    # have to sit inside the same capture
    tb = graphed(lambda: torch.autograd.grad(call(spec, x, fs), [x] + fs, g)) - tf
    bf, bb = 2.0 * (n_in + n_out), 2.0 * (2 * n_in + 2 * n_out)  # fwd: x in, y out; bwd: g + x in (+x again), dx out
    print(f"{spec['tag'][:46]:46s} {spec['count']:4d} | {tf:8.1f} {bf / tf * 1e-3:6.0f} | {tb:8.1f} {bb / tb * 1e-3:6.0f}", flush=True)
    tot_f += tf * spec["count"] * 1e-3
    tot_b += tb * spec["count"] * 1e-3
    tot_bytes += (bf + bb) * spec["count"]
    del x, g, fs
    torch.cuda.empty_cache()
print(f"count-weighted: fwd {tot_f:.2f} ms + bwd {tot_b:.2f} ms = {tot_f + tot_b:.2f} ms per step; "
      f"{tot_bytes * 1e-9:.2f} GB algorithmic -> {tot_bytes / (tot_f + tot_b) * 1e-6:.0f} GB/s")
