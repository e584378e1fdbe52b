"""Host time per adapted layer, forward + backward, Python-driven (what an sd-scripts user pays without whole-step capture):
C++ custom-op dispatch vs the ctypes / Python autograd.Function dispatch vs the reference's own torch call sequence.
Tiny shapes (M = 16): the GPU work is negligible, the wall time between two syncs is host time.
    python benchmarks/host_overhead.py  -> gpurun_out/host_overhead.json"""
import faulthandler
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from lycoris_amd import ops

dev = torch.device("cuda:0")
faulthandler.enable()
faulthandler.dump_traceback_later(40, repeat=True)
N_LAYERS, REPS = 200, 5


def layers(algo):
    out = []
    for _ in range(N_LAYERS):
        x = torch.randn(16, 1280, device=dev, dtype=torch.bfloat16, requires_grad=True)
        g = torch.randn(16, 1280, device=dev, dtype=torch.bfloat16)
        if algo == "lokr":
            fs = [torch.nn.Parameter(torch.randn(8, 8, device=dev)), torch.nn.Parameter(torch.randn(160, 160, device=dev) * 0.05)]
        else:
            fs = [torch.nn.Parameter(torch.randn(16, 1280, device=dev) * 0.05), torch.nn.Parameter(torch.randn(1280, 16, device=dev) * 0.05)]
        W = torch.randn(1280, 1280, device=dev, dtype=torch.bfloat16)
        out.append((x, g, fs, W))
    return out


def native(algo, ls):
    fn = ops.lokr_linear if algo == "lokr" else ops.locon_linear
    ys = [fn(x, fs[0], fs[1], 1.0) for x, g, fs, W in ls]
    torch.autograd.grad(ys, [t for x, g, fs, W in ls for t in [x] + fs], [g for x, g, fs, W in ls])  # one engine call


def native_training(algo, ls):
    """the training configuration: .grad buffers exist, the kernels accumulate straight into them and the weight gradients of
    all layers run in grouped launches (ops.fused_grad_accumulation + the default deferral): only dx comes back from autograd"""
    fn = ops.lokr_linear if algo == "lokr" else ops.locon_linear
    ys = [fn(x, fs[0], fs[1], 1.0) for x, g, fs, W in ls]
    # the factors are listed (a backward call computes only what it is asked for, round 4); the kernels add their gradients in place
    torch.autograd.grad(ys, [t for x, g, fs, W in ls for t in [x] + fs], [g for x, g, fs, W in ls], allow_unused=True)  # one engine call


def native_training_backward(algo, ls):
    """the same through .backward() (every leaf that requires grad is asked for; x.grad is accumulated by autograd as well)"""
    fn = ops.lokr_linear if algo == "lokr" else ops.locon_linear
    ys = [fn(x, fs[0], fs[1], 1.0) for x, g, fs, W in ls]
    torch.autograd.backward(ys, [g for x, g, fs, W in ls])


def native_training_grouped(algo, ls):
    """round 5: sets of three layers reading ONE tensor (to_q / to_k / to_v) through ops.lokr_linear_group -- one dispatch, one autograd
    node and one backward call per set; microseconds per LAYER"""
    ys, wrt = [], []
    for i in range(0, len(ls) - 2, 3):
        x = ls[i][0]
        ys += ops.lokr_linear_group(x, [ls[i + j][2][0] for j in range(3)], [ls[i + j][2][1] for j in range(3)], [1.0, 1.0, 1.0])
        wrt += [x] + [f for j in range(3) for f in ls[i + j][2]]
    n = len(ys)
    torch.autograd.grad(ys, wrt, [ls[k][1] for k in range(n)], allow_unused=True)


def reference(algo, ls):
    ys = []
    for x, g, fs, W in ls:
        dW = torch.kron(fs[0], fs[1]) if algo == "lokr" else fs[1] @ fs[0]
        ys.append(F.linear(x, (W + dW.to(W.dtype)) - W))
    torch.autograd.grad(ys, [t for x, g, fs, W in ls for t in [x] + fs], [g for x, g, fs, W in ls])


def wall(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(REPS):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / N_LAYERS * 1e6


res = {}
for algo in ("lokr", "locon"):
    ls = layers(algo)
    row = {}
    for mode in ("cpp", "python"):
        ops.set_dispatch(mode)
        row[f"native_{mode}_us"] = round(wall(lambda: native(algo, ls)), 1)
    ops.set_dispatch("cpp")
    for x, g, fs, W in ls:
        for f in fs:
            f.grad = torch.zeros_like(f)
    ops.fused_grad_accumulation(True, None)
    try:
        row["native_cpp_training_config_us"] = round(wall(lambda: native_training(algo, ls)), 1)
        row["native_cpp_training_config_backward_us"] = round(wall(lambda: native_training_backward(algo, ls)), 1)
        if algo == "lokr":
            row["native_cpp_training_config_sibling_sets_of_3_us"] = round(wall(lambda: native_training_grouped(algo, ls)) * N_LAYERS / (N_LAYERS // 3 * 3), 1)
    finally:
        ops.fused_grad_accumulation(False, None)
        for x, g, fs, W in ls:
            for f in fs:
                f.grad = None
    row["reference_torch_us"] = round(wall(lambda: reference(algo, ls)), 1)
    res[algo] = row
    print(algo, row, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"what": "host microseconds per adapted layer, fwd + bwd, M = 16 rows (GPU time negligible); native_cpp / native_python / "
                   "reference: plain autograd (factor gradients handed back to the engine); native_cpp_training_config: gradients "
                   "accumulated into existing .grad buffers, weight gradients in grouped launches", "per_layer": res},
          open("gpurun_out/host_overhead.json", "w"), indent=1)
