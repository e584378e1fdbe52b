"""The north-star comparator: the reference's rebuild path through stock PyTorch-ROCm eager ON THE MI355X, next to the
native ops, layer shape by layer shape (development / evidence tool, not the driver's bench.py).

For every distinct SDXL LoKr(factor 8) layer shape: reference call sequence (modules/lokr.py:543-566 -- torch.kron in the
fp32 parameter dtype, cast to the bf16 base-weight dtype, `W + dW - W`, F.linear / F.conv2d, autograd backward), timed
  * eager  : Python loop, wall clock between syncs (what sd-scripts pays per layer), and
  * graph  : the same launches replayed from a hipGraph (kernel time only),
against the native op under the same two regimes.  Count-weighted sums give the adapter part of one training step.

    python benchmarks/eager_baseline.py            -> table + gpurun_out/eager_baseline.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from benchmarks.sdxl_shapes import sdxl_unet_layers
from lycoris_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16
FACTOR = 8


def reference_step(x, W, w1, w2, g, conv):
    f1 = w1.reshape(*w1.shape, *([1] * (w2.dim() - 2)))
    dW = torch.kron(f1, w2.contiguous())
    new_w = W + dW.to(W.dtype)
    delta_w = new_w - W
    y = F.conv2d(x, delta_w, None, conv[0], conv[1]) if conv else F.linear(x, delta_w)
    return torch.autograd.grad(y, [x, w1, w2], g)


def native_step(x, W, w1, w2, g, conv):
    y = ops.lokr_conv2d(x, w1, w2, 1.0, (conv[0],) * 2, (conv[1],) * 2, (1, 1)) if conv else ops.lokr_linear(x, w1, w2, 1.0)
    return torch.autograd.grad(y, [x, w1, w2], g)


def wall(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def graphed(fn, iters):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for _ in range(4):
            fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gph.replay()
    e0.record()
    for _ in range(iters):
        gph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / 4 * 1e3


rows, tot = [], {"ref_eager": 0.0, "ref_graph": 0.0, "nat_eager": 0.0, "nat_graph": 0.0}
for spec in sdxl_unet_layers(1):
    if spec["kind"] == "linear":
        M, I, O = spec["M"], spec["I"], spec["O"]
        x = torch.randn(M, I, device=dev, dtype=dt, requires_grad=True)
        W = torch.randn(O, I, device=dev, dtype=dt)
        w2 = (torch.randn(O // FACTOR, I // FACTOR, device=dev) * 0.05).requires_grad_(True)
        g = torch.randn(M, O, device=dev, dtype=dt)
        conv = None
    else:
        C, O, k = spec["C"], spec["O"], spec["k"]
        x = torch.randn(spec["B"], C, spec["H"], spec["W"], device=dev, dtype=dt, requires_grad=True)
        W = torch.randn(O, C, k, k, device=dev, dtype=dt)
        w2 = (torch.randn(O // FACTOR, C // FACTOR, k, k, device=dev) * 0.05)
        w2 = w2.contiguous(memory_format=torch.channels_last).requires_grad_(True)
        conv = (spec["stride"], spec["pad"])
        ho = (spec["H"] + 2 * spec["pad"] - k) // spec["stride"] + 1
        g = torch.randn(spec["B"], O, ho, ho, device=dev, dtype=dt)
    w1 = (torch.randn(FACTOR, FACTOR, device=dev) * 0.3).requires_grad_(True)
    args = (x, W, w1, w2, g, conv)
    r = {"tag": spec["tag"], "count": spec["count"],
         "ref_eager": wall(lambda: reference_step(*args), 20), "ref_graph": graphed(lambda: reference_step(*args), 10),
         "nat_eager": wall(lambda: native_step(*args), 20), "nat_graph": graphed(lambda: native_step(*args), 10)}
    for k_ in tot:
        tot[k_] += r[k_] * spec["count"] * 1e-3
    rows.append(r)
    print(f"{spec['tag'][:44]:44s} x{spec['count']:3d}  reference eager {r['ref_eager']:8.1f} us  graph {r['ref_graph']:8.1f} us |"
          f" native eager {r['nat_eager']:8.1f} us  graph {r['nat_graph']:8.1f} us | kernel-time ratio {r['ref_graph'] / r['nat_graph']:5.1f}x",
          flush=True)
    del x, W, w1, w2, g
    torch.cuda.empty_cache()
print("adapter fwd+bwd of one SDXL step (788 layers), ms:", {k: round(v, 2) for k, v in tot.items()})
print(f"speed-up over the reference: eager wall {tot['ref_eager'] / tot['nat_eager']:.2f}x, kernel time {tot['ref_graph'] / tot['nat_graph']:.2f}x")
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"rows": rows, "totals_ms": tot}, open("gpurun_out/eager_baseline.json", "w"), indent=1)
