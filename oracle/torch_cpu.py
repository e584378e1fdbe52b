"""CPU baseline: the reference's *rebuild path* call sequence restated with torch CPU ops.  TEST / BENCH
INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg; never imported by the product).

This is what the reference executes per adapted layer on its CPU-runnable path (modules/lokr.py:543-566,
locon.py:309-332, loha.py:301-322, ia3.py:129-144):  materialise dW from the factors, new_weight = W + dW,
delta_weight = new_weight - W, delta = op(x, delta_weight), then autograd backward through the same graph.
/root/reference itself is not available on the GPU box, hence a restatement ("kind": "port").
"""
import time

import torch
import torch.nn.functional as F


def delta_weight(algo, W, factors, scale):
    if algo == "lokr":
        w1, w2 = factors
        f1 = w1.reshape(*w1.shape, *([1] * (w2.dim() - 2)))
        dW = torch.kron(f1, w2.contiguous()) * scale
    elif algo == "locon":
        down, up = factors
        dW = ((up.reshape(up.shape[0], -1) @ down.reshape(down.shape[0], -1)) * scale).reshape(W.shape)
    elif algo == "loha":
        w1a, w1b, w2a, w2b = factors
        dW = (((w1a @ w1b) * (w2a @ w2b)) * scale).reshape(W.shape)
    else:
        raise KeyError(algo)
    new_w = W + dW.to(W.dtype)
    return new_w - W


def time_layer(algo, spec, factors_fn, dtype=torch.float32, reps=2):
    """Seconds for one adapter fwd+bwd (delta only, no base op) of one layer spec on the host CPU."""
    torch.manual_seed(0)
    if spec["kind"] == "linear":
        x = torch.randn(spec["M"], spec["I"], dtype=dtype, requires_grad=True)
        W = torch.randn(spec["O"], spec["I"], dtype=dtype)
        op = lambda x_, w_: F.linear(x_, w_)
    else:
        x = torch.randn(spec["B"], spec["C"], spec["H"], spec["W"], dtype=dtype, requires_grad=True)
        W = torch.randn(spec["O"], spec["C"], spec["k"], spec["k"], dtype=dtype)
        op = lambda x_, w_: F.conv2d(x_, w_, None, spec["stride"], spec["pad"])
    factors = [f.requires_grad_(True) for f in factors_fn(spec)]
    best = float("inf")
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        y = op(x, delta_weight(algo, W, factors, 1.0))
        y.backward(torch.ones_like(y))
        best = min(best, time.perf_counter() - t0)
        x.grad = None
        for f in factors:
            f.grad = None
    return best
