"""Quick per-layer timing of the native ops vs a PyTorch-eager restatement of the reference rebuild path
(lycoris/modules/lokr.py:543-566, locon.py:309-332).  Development tool, not the driver's bench.py."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lycoris_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def eager_lokr(x, w1, w2, W, g):
    dW = torch.kron(w1, w2).to(W.dtype)
    nw = W + dW
    delta_w = nw - W
    y = torch.nn.functional.linear(x, delta_w)
    return torch.autograd.grad(y, [x, w1, w2], g)


def eager_locon(x, down, up, W, g):
    dW = (up @ down).to(W.dtype)
    nw = W + dW
    delta_w = nw - W
    y = torch.nn.functional.linear(x, delta_w)
    return torch.autograd.grad(y, [x, down, up], g)


res = []
for (M, I, O) in [(1024, 1280, 1280), (1024, 1280, 10240), (1024, 5120, 1280), (4096, 640, 640), (4096, 640, 5120), (4096, 2560, 640), (77, 2048, 1280)]:
    x = torch.randn(M, I, device=dev, dtype=dt, requires_grad=True)
    g = torch.randn(M, O, device=dev, dtype=dt)
    W = torch.randn(O, I, device=dev, dtype=dt)
    b, d = 8, I // 8
    a, c = 8, O // 8
    w1 = (torch.randn(a, b, device=dev) * 0.3).requires_grad_(True)
    w2 = (torch.randn(c, d, device=dev) * 0.1).requires_grad_(True)
    down = (torch.randn(16, I, device=dev) * 0.05).requires_grad_(True)
    up = (torch.randn(O, 16, device=dev) * 0.05).requires_grad_(True)

    def nat_lokr_f():
        return ops.lokr_linear(x, w1, w2, 1.0)

    y = nat_lokr_f()

    def nat_lokr_fb():
        y = ops.lokr_linear(x, w1, w2, 1.0)
        torch.autograd.grad(y, [x, w1, w2], g)

    def nat_locon_fb():
        y = ops.locon_linear(x, down, up, 1.0)
        torch.autograd.grad(y, [x, down, up], g)

    def nat_locon_f():
        return ops.locon_linear(x, down, up, 1.0)

    base_f = lambda: torch.nn.functional.linear(x, W)
    r = {
        "shape": (M, I, O),
        "lokr_fwd_us": timeit(nat_lokr_f), "lokr_fwdbwd_us": timeit(nat_lokr_fb),
        "locon_fwd_us": timeit(nat_locon_f), "locon_fwdbwd_us": timeit(nat_locon_fb),
        "eager_lokr_fwdbwd_us": timeit(lambda: eager_lokr(x, w1, w2, W, g)),
        "eager_locon_fwdbwd_us": timeit(lambda: eager_locon(x, down, up, W, g)),
        "base_fwd_us": timeit(base_f),
    }
    e = 2
    r["ideal_us_lokr"] = e * M * (3 * I + 2 * O) / 6.3e12 * 1e6
    print(json.dumps(r), flush=True)
    res.append(r)
json.dump(res, open("gpurun_out/microbench.json", "w"), indent=1)
