"""GPU: deferred, grouped LoKr weight gradients (lyc_lokr_linear_bwd | LYC_DEFER_WGRAD + lyc_lokr_wgrad_group[_ws],
csrc/kron_dw2f.h: full-width tiles, from the kernel arguments or from a problem table in device scratch; csrc/kron_dw2s.h: the
round 1-3 narrow tiles, still used for small factors and selectable with LYC_WGRAD_TILE_S; host side csrc/torch_ops.cpp:
park_deferred / flush_deferred).

The grouped launch computes the same autograd products as the per-layer launches (reference: the factor gradients of
lycoris/modules/lokr.py:543-566), only scheduled together, so the checks are: (a) C ABI: a batch of layers of every tile
configuration, more than one launch's worth of them, a parameter that appears twice, against the oracle and against the
per-layer entry point; (b) custom ops: `.grad` is complete when backward() returns, with any flush threshold, and the
grad-sync callback hears about every parameter exactly once per backward."""
import ctypes

import pytest
import torch
import torch.nn as nn

import oracle
from gpu_util import TOL, err, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

DEFER = 0x200  # LYC_DEFER_WGRAD

# (M, a = b, c, d): SDXL attention (32x32 tiles, 20 slabs when launched alone), the 77-token context, a 640-wide layer,
# a problem on the 64x64 tile configuration (M * a >= 16 k rows), ragged rows / tiny factors, a single row
SHAPES = [(1024, 8, 160, 160), (77, 8, 160, 256), (4096, 8, 80, 80), (4096, 8, 128, 128), (50, 4, 16, 8), (1, 8, 40, 160),
          (333, 16, 24, 40),
          # the full-width tile classes of kron_dw2f.h: 160 x 80 and 80 x 160 (two waves on the rows), several tiles per slab,
          # ragged tiles in both directions, one slab only (the plain store path), a = 16 and a = 4
          (256, 8, 320, 72), (256, 8, 72, 320), (300, 8, 64, 72), (33, 8, 168, 200), (700, 16, 88, 96), (513, 4, 160, 160)]
MODES = {"args": 0, "table": 0, "narrow": 0x400}  # LYC_WGRAD_TILE_S


def _problem(gen, M, a, c, d, dtype):
    """device tensors (g, x, w1, w2) and the float64 numpy arrays of the same (already rounded) values"""
    x, x64 = rnd((M, a * d), dtype, gen)
    g, g64 = rnd((M, a * c), dtype, gen, 0.1)
    w1, w164 = rnd((a, a), torch.float32, gen, 0.3)
    w2, w264 = rnd((c, d), torch.float32, gen, 0.1)
    return (g, x, w1, w2), (g64, x64, w164, w264)


def _oracle(h, alpha):
    g64, x64, w1, w2 = h
    gr = oracle.lokr.backward(x64, g64, w1=w1, w2=w2, scale=alpha)
    return gr["w1"], gr["w2"]


def _per_layer(N, g, x, w1, w2, alpha, code):  # (g, x, w1, w2): device tensors
    M, (a, b), (c, d) = x.shape[0], w1.shape, w2.shape
    dx = torch.empty_like(x)
    dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
    ws = torch.empty(max(int(N.load().lyc_lokr_bwd_workspace_bytes(M, a, b, c, d, code)), 16), dtype=torch.uint8, device=DEV)
    N.call("lyc_lokr_linear_bwd", N.ptr(g), N.ptr(x), N.ptr(w1), N.ptr(w2), N.ptr(dx), N.ptr(dw1), N.ptr(dw2), N.ptr(ws),
           M, a, b, c, d, alpha, code, N.stream_ptr(x.device))
    return dx, dw1, dw2


def _deferred(N, probs, alpha, code, shared=(), mode="args"):
    """dx launches with LYC_DEFER_WGRAD, then ONE lyc_lokr_wgrad_group call over all problems.  `shared`: pairs (i, j) of
    problems whose dw1 / dw2 are the SAME buffers (a module applied twice)."""
    items = (N.WgradItem * len(probs))()
    keep, outs = [], []
    alias = dict(shared)
    for k, (g, x, w1, w2) in enumerate(probs):
        M, (a, b), (c, d) = x.shape[0], w1.shape, w2.shape
        dx = torch.empty_like(x)
        if k in alias:
            dw1, dw2 = outs[alias[k]][1], outs[alias[k]][2]
        else:
            dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
        ws = torch.empty(max(int(N.load().lyc_lokr_bwd_workspace_bytes(M, a, b, c, d, code)), 16), dtype=torch.uint8, device=DEV)
        assert N.load().lyc_lokr_wgrad_deferrable(N.ptr(g), N.ptr(x), M, a, b, c, d, code) == 1
        N.call("lyc_lokr_linear_bwd", N.ptr(g), N.ptr(x), N.ptr(w1), N.ptr(w2), N.ptr(dx), N.ptr(dw1), None, N.ptr(ws),
               M, a, b, c, d, alpha, code | DEFER, N.stream_ptr(x.device))
        items[k] = N.WgradItem(N.ptr(g), N.ptr(x), N.ptr(w1), N.ptr(dw1), N.ptr(dw2), N.ptr(ws), M, a, b, c, d, alpha)
        keep.append(ws)
        outs.append((dx, dw1, dw2))
    if mode == "table":
        tb = int(N.load().lyc_lokr_wgrad_table_bytes(len(probs)))
        table = torch.empty(tb, dtype=torch.uint8, device=DEV)
        N.call("lyc_lokr_wgrad_group_ws", ctypes.cast(items, ctypes.c_void_p), len(probs), code, N.ptr(table), tb,
               N.stream_ptr(probs[0][1].device))
    else:
        N.call("lyc_lokr_wgrad_group", ctypes.cast(items, ctypes.c_void_p), len(probs), code | MODES[mode],
               N.stream_ptr(probs[0][1].device))
    torch.cuda.synchronize()
    return outs


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_grouped_weight_gradients_match_the_oracle_and_the_per_layer_path(dtype, mode):
    from lycoris_amd import _native as N
    gen = torch.Generator().manual_seed(11)
    code = N.dtype_code(dtype)
    alpha = 0.7
    both = [_problem(gen, *s, dtype) for s in SHAPES]
    both += [_problem(gen, 64, 8, 16, 16, dtype) for _ in range(30)]  # > 24 items of one configuration: two launches
    both += [_problem(gen, 40, 8, 64, 64, dtype) for _ in range(27)]  # the same on the full-width tiles
    probs, host = [b[0] for b in both], [b[1] for b in both]
    got = _deferred(N, probs, alpha, code, mode=mode)
    bound = TOL["f32_out"][dtype]
    for k, (h, (dx, dw1, dw2)) in enumerate(zip(host, got)):
        r1, r2 = _oracle(h, alpha)
        assert err(dw1, r1) <= bound, (k, "dw1", err(dw1, r1))
        assert err(dw2, r2) <= bound, (k, "dw2", err(dw2, r2))
        if k < len(SHAPES):  # dx is the same launch with or without the flag
            px, p1, p2 = _per_layer(N, *probs[k], alpha, code)
            assert torch.equal(dx, px), k
            assert err(dw1, p1.double().cpu().numpy()) <= 1e-5 and err(dw2, p2.double().cpu().numpy()) <= 1e-5, k


@pytest.mark.parametrize("mode,cd", [("args", 32), ("args", 160), ("table", 160), ("narrow", 160)])
def test_a_parameter_that_appears_twice_in_one_group_is_added_atomically(mode, cd):
    from lycoris_amd import _native as N
    gen = torch.Generator().manual_seed(12)
    dtype, alpha = torch.bfloat16, 1.0
    code = N.dtype_code(dtype)
    (ga, xa, w1, w2), ha = _problem(gen, 64, 8, cd, cd, dtype)   # one slab each: the plain (non-atomic) store path when alone
    (gb, xb, _, _), hb = _problem(gen, 64, 8, cd, cd, dtype)
    hb = (hb[0], hb[1], ha[2], ha[3])
    got = _deferred(N, [(ga, xa, w1, w2), (gb, xb, w1, w2)], alpha, code, shared=[(1, 0)], mode=mode)
    a1, a2 = _oracle(ha, alpha)
    b1, b2 = _oracle(hb, alpha)
    assert got[0][1].data_ptr() == got[1][1].data_ptr()
    bound = TOL["f32_out"][dtype]
    assert err(got[0][1], a1 + b1) <= bound and err(got[0][2], a2 + b2) <= bound


def test_group_call_rejects_what_it_cannot_run():
    from lycoris_amd import _native as N
    x = torch.zeros(8, 24, device=DEV, dtype=torch.float32)
    assert N.load().lyc_lokr_wgrad_deferrable(N.ptr(x), N.ptr(x), 8, 4, 4, 6, 6, N.LYC_F32) == 0   # fp32 activations
    assert N.load().lyc_lokr_wgrad_deferrable(N.ptr(x), N.ptr(x), 8, 4, 4, 6, 6, N.LYC_BF16) == 0  # c, d not multiples of 8
    items = (N.WgradItem * 1)()
    items[0] = N.WgradItem(N.ptr(x), N.ptr(x), N.ptr(x), None, N.ptr(x), None, 8, 4, 4, 6, 6, 1.0)
    with pytest.raises(RuntimeError, match="fast path"):
        N.call("lyc_lokr_wgrad_group", ctypes.cast(items, ctypes.c_void_p), 1, N.LYC_BF16, None)
    N.call("lyc_lokr_wgrad_group", None, 0, N.LYC_BF16, None)  # an empty batch is fine


class _Stack(nn.Module):
    """a few LoKr-adapted layers in sequence, one of them applied twice (weight sharing inside one backward)"""

    def __init__(self, n=5, a=8, c=16, d=16):
        super().__init__()
        self.w1 = nn.ParameterList([nn.Parameter(torch.randn(a, a, device=DEV) * 0.3) for _ in range(n)])
        self.w2 = nn.ParameterList([nn.Parameter(torch.randn(c, d, device=DEV) * 0.1) for _ in range(n)])

    def forward(self, x):
        from lycoris_amd import ops
        order = list(range(len(self.w1))) + [1]
        for i in order:
            x = x + ops.lokr_linear(x, self.w1[i], self.w2[i], 0.5)
        return x


@pytest.mark.parametrize("flush_at", [48, 2], ids=["end_of_backward", "every_2_layers"])
def test_grads_are_complete_when_backward_returns_and_every_parameter_is_reported(flush_at):
    from lycoris_amd import ops
    torch.manual_seed(5)
    net = _Stack()
    x = (torch.randn(96, 128, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    gy = (torch.randn(96, 128, device=DEV) * 0.1).to(torch.bfloat16)
    params = list(net.parameters())

    def run(defer):
        seen = []
        for p in params:
            p.grad = torch.zeros_like(p)
        x.grad = None
        ops.fused_grad_accumulation(True, callback=lambda p: seen.append(id(p)))
        ops.deferred_weight_gradients(defer, flush_at)
        try:
            net(x).backward(gy)
            assert ops._DISPATCH["ext"].deferred_pending() == 0  # nothing parked once backward() has returned
            torch.cuda.synchronize()
            return x.grad.clone(), [p.grad.clone() for p in params], seen
        finally:
            ops.fused_grad_accumulation(False, None)
            ops.deferred_weight_gradients(True, 48)

    dx0, g0, seen0 = run(False)
    dx1, g1, seen1 = run(True)
    assert torch.equal(dx0, dx1)
    for p, u, v in zip(params, g0, g1):
        assert float(u.abs().max()) > 0
        assert torch.allclose(u, v, rtol=2e-4, atol=1e-6), float((u - v).abs().max())
    # 6 layer calls on 5 layers: every parameter is reported ONCE per backward pass, after its last accumulation -- the shared
    # layer's pair too (the contract of an autograd post-accumulate hook; round 3), in both modes
    assert sorted(seen0) == sorted(seen1) == sorted(id(p) for p in params) and len(seen1) == 10


def test_parameters_without_a_grad_buffer_are_not_deferred():
    """plain autograd (gradients handed back to the engine) keeps the per-layer launches"""
    from lycoris_amd import ops
    net = _Stack(n=2)
    x = (torch.randn(32, 128, device=DEV)).to(torch.bfloat16).requires_grad_(True)
    ops.fused_grad_accumulation(True, None)
    try:
        y = net(x)
        grads = torch.autograd.grad(y, list(net.parameters()), torch.ones_like(y) * 0.01)
        assert all(g is not None and float(g.abs().max()) > 0 for g in grads)
        assert ops._DISPATCH["ext"].deferred_pending() == 0
    finally:
        ops.fused_grad_accumulation(False, None)


# ---- LoCon: lyc_locon_linear_bwd with d_down = d_up = NULL, then lyc_locon_wgrad_group --------------------------------------
LOCON_SHAPES = [(1024, 1280, 1280, 16), (77, 2048, 640, 16), (4096, 640, 640, 8), (50, 72, 40, 4), (1, 1280, 320, 16),
                (300, 96, 136, 24), (128, 64, 64, 40)]  # (M, I, O, r): ranks on all three rank tiles, odd widths


def _locon_problem(gen, M, I, O, r, dtype):
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 0.1)
    down, d64 = rnd((r, I), torch.float32, gen, 0.1)
    up, u64 = rnd((O, r), torch.float32, gen, 0.1)
    return (g, x, down, up), (g64, x64, d64, u64)


def _locon_deferred(N, probs, alpha, code, shared=()):
    items = (N.LoconWgradItem * len(probs))()
    outs, keep = [], []
    alias = dict(shared)
    for k, (g, x, down, up) in enumerate(probs):
        M, I, O, r = x.shape[0], x.shape[1], g.shape[1], down.shape[0]
        t = torch.empty(M, r, device=DEV)
        y = torch.empty_like(g)
        N.call("lyc_locon_linear_fwd", N.ptr(x), N.ptr(down), N.ptr(up), N.ptr(t), N.ptr(y), M, I, O, r, alpha, code, N.stream_ptr(x.device))
        dt, dx = torch.empty(M, r, device=DEV), torch.empty_like(x)
        dd, du = (outs[alias[k]][1], outs[alias[k]][2]) if k in alias else (torch.zeros_like(down), torch.zeros_like(up))
        assert N.load().lyc_locon_wgrad_deferrable(N.ptr(g), N.ptr(x), M, I, O, r, code) == 1
        N.call("lyc_locon_linear_bwd", N.ptr(g), N.ptr(x), N.ptr(down), N.ptr(up), N.ptr(t), N.ptr(dt), N.ptr(dx), None, None,
               M, I, O, r, alpha, code, N.stream_ptr(x.device))
        items[k] = N.LoconWgradItem(N.ptr(g), N.ptr(x), N.ptr(t), N.ptr(dt), N.ptr(dd), N.ptr(du), M, I, O, r, alpha)
        keep += [t, dt]
        outs.append((dx, dd, du))
    N.call("lyc_locon_wgrad_group", ctypes.cast(items, ctypes.c_void_p), len(probs), code, N.stream_ptr(probs[0][1].device))
    torch.cuda.synchronize()
    return outs


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_grouped_locon_factor_gradients_match_the_oracle(dtype):
    from lycoris_amd import _native as N
    gen = torch.Generator().manual_seed(21)
    code = N.dtype_code(dtype)
    alpha = 0.7
    both = [_locon_problem(gen, *s, dtype) for s in LOCON_SHAPES]
    both += [_locon_problem(gen, 64, 128, 128, 16, dtype) for _ in range(22)]  # > 18 items of one configuration: two launches
    probs, host = [b[0] for b in both], [b[1] for b in both]
    got = _locon_deferred(N, probs, alpha, code)
    bound = TOL["f32_out"][dtype]
    for k, ((g64, x64, d64, u64), (dx, dd, du)) in enumerate(zip(host, got)):
        rx, rd, ru = oracle.locon.backward(x64, g64, d64, u64, scale=alpha)
        assert err(dd, rd) <= bound, (k, "d_down", err(dd, rd))
        assert err(du, ru) <= bound, (k, "d_up", err(du, ru))
        assert err(dx, rx, dtype) <= TOL["store_out"][dtype], (k, "dx")


def test_a_shared_locon_layer_in_one_group_is_added_atomically():
    from lycoris_amd import _native as N
    gen = torch.Generator().manual_seed(22)
    dtype, alpha = torch.bfloat16, 1.0
    code = N.dtype_code(dtype)
    (ga, xa, down, up), ha = _locon_problem(gen, 96, 128, 64, 8, dtype)  # one slab each: plain stores when alone
    (gb, xb, _, _), hb = _locon_problem(gen, 96, 128, 64, 8, dtype)
    got = _locon_deferred(N, [(ga, xa, down, up), (gb, xb, down, up)], alpha, code, shared=[(1, 0)])
    _, ad, au = oracle.locon.backward(ha[1], ha[0], ha[2], ha[3], scale=alpha)
    _, bd, bu = oracle.locon.backward(hb[1], hb[0], ha[2], ha[3], scale=alpha)
    bound = TOL["f32_out"][dtype]
    assert err(got[0][1], ad + bd) <= bound and err(got[0][2], au + bu) <= bound


class _LoconStack(nn.Module):
    def __init__(self, n=4, width=128, r=16):
        super().__init__()
        self.down = nn.ParameterList([nn.Parameter(torch.randn(r, width, device=DEV) * 0.1) for _ in range(n)])
        self.up = nn.ParameterList([nn.Parameter(torch.randn(width, r, device=DEV) * 0.1) for _ in range(n)])

    def forward(self, x):
        from lycoris_amd import ops
        for i in list(range(len(self.down))) + [0]:
            x = x + ops.locon_linear(x, self.down[i], self.up[i], 0.5)
        return x


@pytest.mark.parametrize("flush_at", [48, 3], ids=["end_of_backward", "every_3_layers"])
def test_locon_grads_are_complete_when_backward_returns(flush_at):
    from lycoris_amd import ops
    torch.manual_seed(6)
    net = _LoconStack()
    x = (torch.randn(80, 128, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    gy = (torch.randn(80, 128, device=DEV) * 0.1).to(torch.bfloat16)
    params = list(net.parameters())

    def run(defer):
        seen = []
        for p in params:
            p.grad = torch.zeros_like(p)
        x.grad = None
        ops.fused_grad_accumulation(True, callback=lambda p: seen.append(id(p)))
        ops.deferred_weight_gradients(defer, flush_at)
        try:
            net(x).backward(gy)
            assert ops._DISPATCH["ext"].deferred_pending() == 0
            torch.cuda.synchronize()
            return x.grad.clone(), [p.grad.clone() for p in params], seen
        finally:
            ops.fused_grad_accumulation(False, None)
            ops.deferred_weight_gradients(True, 48)

    dx0, g0, seen0 = run(False)
    dx1, g1, seen1 = run(True)
    assert torch.equal(dx0, dx1)
    for u, v in zip(g0, g1):
        assert float(u.abs().max()) > 0 and torch.allclose(u, v, rtol=2e-4, atol=1e-6), float((u - v).abs().max())
    assert sorted(seen0) == sorted(seen1) == sorted(id(p) for p in params) and len(seen1) == 8  # once per parameter


# ---- LoHa: lyc_loha_linear_bwd with NULL gradient pointers (dx only), then lyc_loha_wgrad_group -----------------------------
LOHA_SHAPES = [(256, 1280, 1280, 32), (77, 2048, 640, 32), (300, 640, 2560, 16), (50, 72, 40, 4), (1, 320, 1280, 32),
               (128, 200, 136, 40)]  # (M, I, O, r): blocks of 4 x n, 2 x n and single tiles, ragged edges, a rank > 32


def _loha_problem(gen, M, I, O, r, dtype):
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 0.1)
    fs = [rnd((O, r), torch.float32, gen, 0.3), rnd((r, I), torch.float32, gen, 0.3), rnd((O, r), torch.float32, gen, 0.3),
          rnd((r, I), torch.float32, gen, 0.3)]
    return (g, x, [f[0] for f in fs]), (g64, x64, [f[1] for f in fs])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_grouped_loha_factor_gradients_match_the_oracle(dtype):
    from lycoris_amd import _native as N
    gen = torch.Generator().manual_seed(31)
    code = N.dtype_code(dtype)
    alpha = 0.7
    both = [_loha_problem(gen, *s, dtype) for s in LOHA_SHAPES]
    both += [_loha_problem(gen, 64, 128, 128, 8, dtype) for _ in range(26)]  # > 24 items of one configuration
    items = (N.LohaWgradItem * len(both))()
    outs, keep = [], []
    for k, ((g, x, fs), _) in enumerate(both):
        M, I, O, r = x.shape[0], x.shape[1], g.shape[1], fs[0].shape[1]
        ws = torch.empty(int(N.load().lyc_loha_workspace_bytes(O, I, code)), dtype=torch.uint8, device=DEV)
        y, dx = torch.empty_like(g), torch.empty_like(x)
        N.call("lyc_loha_linear_fwd", N.ptr(x), *[N.ptr(f) for f in fs], N.ptr(ws), N.ptr(y), M, I, O, r, alpha, code, N.stream_ptr(x.device))
        N.call("lyc_loha_linear_bwd", N.ptr(g), N.ptr(x), *[N.ptr(f) for f in fs], N.ptr(ws), None, N.ptr(dx), None, None, None, None,
               M, I, O, r, alpha, code, N.stream_ptr(x.device))
        ds = [torch.zeros_like(f) for f in fs]
        gw = torch.empty(O, I, device=DEV)
        items[k] = N.LohaWgradItem(N.ptr(g), N.ptr(x), *[N.ptr(f) for f in fs], *[N.ptr(d) for d in ds], N.ptr(gw), M, I, O, r, alpha)
        keep += [ws, gw]
        outs.append((dx, ds))
    N.call("lyc_loha_wgrad_group", ctypes.cast(items, ctypes.c_void_p), len(both), code, N.stream_ptr(both[0][0][1].device))
    torch.cuda.synchronize()
    for k, ((_, (g64, x64, f64)), (dx, ds)) in enumerate(zip(both, outs)):
        rx, *rf = oracle.loha.backward(x64, g64, *f64, scale=alpha)
        for name, d, want in zip(("w1a", "w1b", "w2a", "w2b"), ds, rf):
            assert err(d, want) <= TOL["f32_out"][dtype], (k, name, err(d, want))
        assert err(dx, rx, dtype) <= TOL["loha_store"][dtype], (k, "dx")
        rx_cast = oracle.loha.backward(x64, g64, *f64, scale=alpha, round_dw=str(dtype))[0]  # modules/loha.py:310
        assert err(dx, rx_cast, dtype) <= TOL["store_out"][dtype], (k, "dx@cast", err(dx, rx_cast, dtype))


class _LohaStack(nn.Module):
    def __init__(self, n=3, width=128, r=8):
        super().__init__()
        mk = lambda *s: nn.Parameter(torch.randn(*s, device=DEV) * 0.3)
        self.f = nn.ModuleList([nn.ParameterList([mk(width, r), mk(r, width), mk(width, r), mk(r, width)]) for _ in range(n)])

    def forward(self, x):
        from lycoris_amd import ops
        for i in list(range(len(self.f))) + [0]:
            x = x + ops.loha_linear(x, *self.f[i], 0.5)
        return x


def test_loha_grads_are_complete_when_backward_returns():
    from lycoris_amd import ops
    torch.manual_seed(7)
    net = _LohaStack()
    x = (torch.randn(80, 128, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    gy = (torch.randn(80, 128, device=DEV) * 0.1).to(torch.bfloat16)
    params = list(net.parameters())

    def run(defer):
        seen = []
        for p in params:
            p.grad = torch.zeros_like(p)
        x.grad = None
        ops.fused_grad_accumulation(True, callback=lambda p: seen.append(id(p)))
        ops.deferred_weight_gradients(defer, 48)
        try:
            net(x).backward(gy)
            assert ops._DISPATCH["ext"].deferred_pending() == 0
            torch.cuda.synchronize()
            return x.grad.clone(), [p.grad.clone() for p in params], seen
        finally:
            ops.fused_grad_accumulation(False, None)
            ops.deferred_weight_gradients(True, 48)

    dx0, g0, seen0 = run(False)
    dx1, g1, seen1 = run(True)
    assert torch.equal(dx0, dx1)
    for u, v in zip(g0, g1):
        assert float(u.abs().max()) > 0 and torch.allclose(u, v, rtol=2e-4, atol=1e-6), float((u - v).abs().max())
    assert sorted(seen0) == sorted(seen1) == sorted(id(p) for p in params) and len(seen1) == 12  # once per parameter


def test_small_batches_take_the_single_launch_plans():
    """fewer than 8 layers in a call: planned like per-layer launches (the batch plans would starve the chip)"""
    from lycoris_amd import _native as N
    gen = torch.Generator().manual_seed(41)
    dtype, alpha = torch.bfloat16, 0.7
    code = N.dtype_code(dtype)
    both = [_problem(gen, *s, dtype) for s in SHAPES[:3]]
    got = _deferred(N, [b[0] for b in both], alpha, code)
    for k, ((_, h), (dx, dw1, dw2)) in enumerate(zip(both, got)):
        r1, r2 = _oracle(h, alpha)
        assert err(dw1, r1) <= TOL["f32_out"][dtype] and err(dw2, r2) <= TOL["f32_out"][dtype], k
    lboth = [_locon_problem(gen, *s, dtype) for s in LOCON_SHAPES[:3]]
    lgot = _locon_deferred(N, [b[0] for b in lboth], alpha, code)
    for k, ((_, (g64, x64, d64, u64)), (dx, dd, du)) in enumerate(zip(lboth, lgot)):
        _, rd, ru = oracle.locon.backward(x64, g64, d64, u64, scale=alpha)
        assert err(dd, rd) <= TOL["f32_out"][dtype] and err(du, ru) <= TOL["f32_out"][dtype], k


def test_deferred_gradients_under_activation_checkpointing():
    """torch.utils.checkpoint re-runs the forward inside the backward pass: the parked layers of the re-run segment are flushed
    by the same end-of-backward callback"""
    from torch.utils.checkpoint import checkpoint
    from lycoris_amd import ops
    torch.manual_seed(8)
    net = _Stack(n=3)
    x = (torch.randn(64, 128, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    gy = (torch.randn(64, 128, device=DEV) * 0.1).to(torch.bfloat16)
    params = list(net.parameters())

    def run(ckpt):
        for p in params:
            p.grad = torch.zeros_like(p)
        x.grad = None
        ops.fused_grad_accumulation(True, None)
        try:
            y = checkpoint(net, x, use_reentrant=False) if ckpt else net(x)
            y.backward(gy)
            assert ops._DISPATCH["ext"].deferred_pending() == 0
            torch.cuda.synchronize()
            return x.grad.clone(), [p.grad.clone() for p in params]
        finally:
            ops.fused_grad_accumulation(False, None)

    dx0, g0 = run(False)
    dx1, g1 = run(True)
    assert torch.equal(dx0, dx1)
    for u, v in zip(g0, g1):
        assert torch.allclose(u, v, rtol=2e-4, atol=1e-6)


def test_parked_layers_of_a_failed_backward_can_be_discarded():
    from lycoris_amd import ops
    net = _Stack(n=2)
    x = (torch.randn(32, 128, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    ops.fused_grad_accumulation(True, None)
    try:
        y = net(x)

        def boom(g):
            raise RuntimeError("boom")

        x.register_hook(boom)  # the last thing the backward pass does: both layers are parked by then
        with pytest.raises(RuntimeError, match="boom"):
            y.backward(torch.ones_like(y))
        ops.discard_deferred()  # (whether the engine ran its final callbacks after the error is an implementation detail)
        assert ops._DISPATCH["ext"].deferred_pending() == 0
        # the next pass is complete again: its own end-of-backward callback is installed although the failed pass never ran its
        x2 = x.detach().clone().requires_grad_(True)
        want = []
        for defer in (False, True):
            ops.deferred_weight_gradients(defer)
            for p in net.parameters():
                p.grad = torch.zeros_like(p)
            net(x2).backward(torch.ones_like(y) * 0.01)
            assert ops._DISPATCH["ext"].deferred_pending() == 0
            torch.cuda.synchronize()
            want.append([p.grad.clone() for p in net.parameters()])
        for u, v in zip(*want):
            assert float(u.abs().max()) > 0 and torch.allclose(u, v, rtol=2e-4, atol=1e-6)
    finally:
        ops.deferred_weight_gradients(True)
        ops.fused_grad_accumulation(False, None)
        ops.discard_deferred()
