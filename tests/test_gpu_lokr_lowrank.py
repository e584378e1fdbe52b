"""GPU: LoKr with a low-rank second factor, w2 = w2_a @ w2_b (reference lycoris/modules/lokr.py:131-136 make the two factors,
:370 forms the product, functional/lokr.py:124-151 the same in the functional API; BASELINE configs[3] "(low)").

The native path never forms the product as a tensor: `lyc_lokr_pack_group` packs the kernels' operand planes from the two
factors, the weight gradient dW2 lands in a scratch, `lyc_lokr_lr_chain_group` (csrc/kron_conv.h: kron_lr_chain_kernel) applies
d_w2a += dW2 w2b^T, d_w2b += w2a^T dW2 for a batch of layers.  Checked here: (a) the chain kernel alone against float64;
(b) the custom op `lokr_linear_lr` against the numpy oracle -- gradients handed back to autograd, accumulated into `.grad`,
parked for the grouped launch -- and against the op on the materialised product; (c) shapes off the fast path; (d) the module."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
from gpu_util import TOL, err, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT16 = pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])


# ---- (a) the chain rule of the product, grouped ---------------------------------------------------------------------------
def test_chain_group_matches_float64():
    from lycoris_amd import _native as N
    gen = torch.Generator().manual_seed(31)
    shapes = [(160, 160, 16), (80, 320, 4), (8, 8, 1), (40, 24, 7), (1280, 160, 16), (16, 1280, 32)] + [(32, 32, 8)] * 60  # > one launch
    items = (N.LokrLrChainItem * len(shapes))()
    keep, want = [], []
    for k, (c, d, r) in enumerate(shapes):
        dw2, dw264 = rnd((c, d), torch.float32, gen)
        a, a64 = rnd((c, r), torch.float32, gen, 0.3)
        b, b64 = rnd((r, d), torch.float32, gen, 0.3)
        da0, da064 = rnd((c, r), torch.float32, gen)  # accumulated INTO: existing content stays
        db0, db064 = rnd((r, d), torch.float32, gen)
        items[k] = N.LokrLrChainItem(N.ptr(dw2), N.ptr(a), N.ptr(b), N.ptr(da0), N.ptr(db0), c, d, r)
        keep.append((dw2, a, b, da0, db0))
        want.append((da064 + dw264 @ b64.T, db064 + a64.T @ dw264))
    N.call("lyc_lokr_lr_chain_group", ctypes.cast(items, ctypes.c_void_p), len(shapes), N.stream_ptr(torch.device(DEV)))
    torch.cuda.synchronize()
    for k, ((_, _, _, da, db), (wa, wb)) in enumerate(zip(keep, want)):
        assert err(da, wa) <= 2e-6, (k, shapes[k], err(da, wa))
        assert err(db, wb) <= 2e-6, (k, shapes[k], err(db, wb))
    N.call("lyc_lokr_lr_chain_group", None, 0, None)  # an empty batch is fine


def test_chain_group_one_factor_only_and_shared_targets():
    """d_w2a or d_w2b may be NULL (a frozen factor); two items may add into the same buffers (a module applied twice)"""
    from lycoris_amd import _native as N
    gen = torch.Generator().manual_seed(32)
    c, d, r = 48, 40, 8
    a, a64 = rnd((c, r), torch.float32, gen, 0.3)
    b, b64 = rnd((r, d), torch.float32, gen, 0.3)
    g1, g164 = rnd((c, d), torch.float32, gen)
    g2, g264 = rnd((c, d), torch.float32, gen)
    da, db = torch.zeros(c, r, device=DEV), torch.zeros(r, d, device=DEV)
    items = (N.LokrLrChainItem * 3)()
    items[0] = N.LokrLrChainItem(N.ptr(g1), N.ptr(a), N.ptr(b), N.ptr(da), N.ptr(db), c, d, r)
    items[1] = N.LokrLrChainItem(N.ptr(g2), N.ptr(a), N.ptr(b), N.ptr(da), N.ptr(db), c, d, r)
    only_b = torch.zeros(r, d, device=DEV)
    items[2] = N.LokrLrChainItem(N.ptr(g1), N.ptr(a), N.ptr(b), None, N.ptr(only_b), c, d, r)
    N.call("lyc_lokr_lr_chain_group", ctypes.cast(items, ctypes.c_void_p), 3, N.stream_ptr(torch.device(DEV)))
    torch.cuda.synchronize()
    assert err(da, (g164 + g264) @ b64.T) <= 2e-6 and err(db, a64.T @ (g164 + g264)) <= 2e-6
    assert err(only_b, a64.T @ g164) <= 2e-6


# ---- (b) the custom op --------------------------------------------------------------------------------------------------------
# (M, a, c, d, r): SDXL attention / feed-forward at rank 16, the 77-token context, ragged rows, a small factor with rank 4
OP_SHAPES = [(1024, 8, 160, 160, 16), (256, 8, 160, 640, 16), (77, 8, 160, 256, 16), (333, 8, 40, 24, 4), (64, 4, 32, 48, 8), (1, 8, 80, 80, 16)]


def _problem(gen, M, a, c, d, r, dtype):
    x, x64 = rnd((M, a * d), dtype, gen)
    g, g64 = rnd((M, a * c), dtype, gen, 0.1)
    w1, w164 = rnd((a, a), torch.float32, gen, 0.3)
    w2a, a64 = rnd((c, r), torch.float32, gen, 0.3)
    w2b, b64 = rnd((r, d), torch.float32, gen, 0.3)
    return (x, g, w1, w2a, w2b), (x64, g64, w164, a64, b64)


def _oracle(h, alpha):
    x64, g64, w1, a, b = h
    y = oracle.lokr.forward(x64, w1=w1, w2a=a, w2b=b, scale=alpha)
    gr = oracle.lokr.backward(x64, g64, w1=w1, w2a=a, w2b=b, scale=alpha)
    return y, gr


@DT16
@pytest.mark.parametrize("shape", OP_SHAPES, ids=[f"M{s[0]}_a{s[1]}_c{s[2]}_d{s[3]}_r{s[4]}" for s in OP_SHAPES])
def test_op_matches_the_oracle_gradients_handed_back(shape, dtype):
    from lycoris_amd import ops
    gen = torch.Generator().manual_seed(sum(shape))
    (x, g, w1, w2a, w2b), h = _problem(gen, *shape, dtype)
    alpha = 0.7
    x.requires_grad_(True)
    ps = [nn.Parameter(t) for t in (w1, w2a, w2b)]
    y = ops.lokr_linear_lr(x, ps[0], ps[1], ps[2], alpha)
    dx, d1, da, db = torch.autograd.grad(y, [x] + ps, g)
    torch.cuda.synchronize()
    y_ref, gr = _oracle(h, alpha)
    st, f32 = TOL["store_out"][dtype], TOL["f32_out"][dtype]
    assert err(y, y_ref, dtype) <= st, err(y, y_ref, dtype)
    assert err(dx, gr["dx"], dtype) <= st, err(dx, gr["dx"], dtype)
    assert err(d1, gr["w1"]) <= f32, err(d1, gr["w1"])
    assert err(da, gr["w2a"]) <= f32, err(da, gr["w2a"])
    assert err(db, gr["w2b"]) <= f32, err(db, gr["w2b"])
    # the same layer through the materialised product: same kernels behind different operand sources
    y2 = ops.lokr_linear(x, ps[0], ps[1] @ ps[2], alpha)
    assert float((y.detach().float() - y2.detach().float()).norm() / y2.detach().float().norm()) <= 2e-3


class _Stack(nn.Module):
    """low-rank LoKr layers in sequence, one applied twice, one full-matrix layer in between (mixed parked lists)"""

    def __init__(self, n=4, a=8, c=16, d=16, r=4):
        super().__init__()
        mk = lambda *s, sc: nn.Parameter(torch.randn(*s, device=DEV) * sc)
        self.w1 = nn.ParameterList([mk(a, a, sc=0.3) for _ in range(n)])
        self.w2a = nn.ParameterList([mk(c, r, sc=0.3) for _ in range(n)])
        self.w2b = nn.ParameterList([mk(r, d, sc=0.3) for _ in range(n)])
        self.full1, self.full2 = mk(a, a, sc=0.3), mk(c, d, sc=0.1)

    def forward(self, x, materialise=False):
        from lycoris_amd import ops
        for i in list(range(len(self.w1))) + [1]:
            if materialise:
                x = x + ops.lokr_linear(x, self.w1[i], self.w2a[i] @ self.w2b[i], 0.5)
            else:
                x = x + ops.lokr_linear_lr(x, self.w1[i], self.w2a[i], self.w2b[i], 0.5)
            if i == 2:
                x = x + ops.lokr_linear(x, self.full1, self.full2, 0.5)
        return x


@pytest.mark.parametrize("flush_at", [48, 2], ids=["end_of_backward", "every_2_layers"])
def test_parked_low_rank_layers_grads_complete_and_reported_once(flush_at):
    from lycoris_amd import ops
    torch.manual_seed(6)
    net = _Stack()
    x = (torch.randn(96, 128, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    gy = (torch.randn(96, 128, device=DEV) * 0.1).to(torch.bfloat16)
    params = list(net.parameters())
    # truth: autograd through the materialised product, gradients handed back
    want = torch.autograd.grad(net(x, materialise=True), params, gy)

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
    for p, u, v, w in zip(params, g0, g1, want):
        assert float(u.abs().max()) > 0
        assert float((u - v).norm() / u.norm()) <= 2e-5
        assert float((v - w).norm() / w.norm()) <= 2e-4, float((v - w).norm() / w.norm())
    assert sorted(seen0) == sorted(seen1) == sorted(id(p) for p in params)


def test_second_step_sees_the_updated_factors():
    """the planes are cached per (w2a, w2b) pair and repacked when either factor changed (optimizer.step() bumps the version)"""
    from lycoris_amd import ops
    gen = torch.Generator().manual_seed(33)
    (x, g, w1, w2a, w2b), _ = _problem(gen, 128, 8, 32, 32, 8, torch.bfloat16)
    pa, pb = nn.Parameter(w2a), nn.Parameter(w2b)
    y0 = ops.lokr_linear_lr(x, w1, pa, pb, 1.0).clone()
    with torch.no_grad():
        pb.mul_(2.0)
    y1 = ops.lokr_linear_lr(x, w1, pa, pb, 1.0).clone()
    with torch.no_grad():
        pa.mul_(0.25)
    y2 = ops.lokr_linear_lr(x, w1, pa, pb, 1.0)
    torch.cuda.synchronize()
    assert float((y1.float() - 2 * y0.float()).norm() / y0.float().norm()) <= 1e-2
    assert float((y2.float() - 0.5 * y0.float()).norm() / y0.float().norm()) <= 1e-2


def test_base_is_added_in_the_epilogue():
    from lycoris_amd import ops
    gen = torch.Generator().manual_seed(34)
    (x, g, w1, w2a, w2b), _ = _problem(gen, 256, 8, 40, 40, 8, torch.bfloat16)
    base = (torch.randn(256, 320, device=DEV)).to(torch.bfloat16)
    d = ops.lokr_linear_lr(x, w1, w2a, w2b, 1.0)
    y = ops.lokr_linear_lr(x, w1, w2a, w2b, 1.0, base=base)
    torch.cuda.synchronize()
    want = (base.float() + d.float())
    assert float((y.float() - want).norm() / want.norm()) <= 4e-3  # one bf16 rounding of the sum vs two


# ---- (c) shapes the plane kernels do not take: the product is formed and the row kernels run ---------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(50, 4, 6, 10, 3), (33, 3, 5, 7, 2)], ids=["c6_d10", "a3_c5_d7"])
def test_off_fast_path_shapes(shape, dtype):
    from lycoris_amd import ops
    gen = torch.Generator().manual_seed(sum(shape) + 1)
    (x, g, w1, w2a, w2b), h = _problem(gen, *shape, dtype)
    x.requires_grad_(True)
    ps = [nn.Parameter(t) for t in (w1, w2a, w2b)]
    y = ops.lokr_linear_lr(x, ps[0], ps[1], ps[2], 1.0)
    dx, d1, da, db = torch.autograd.grad(y, [x] + ps, g)
    torch.cuda.synchronize()
    y_ref, gr = _oracle(h, 1.0)
    st, f32 = TOL["store_out"][dtype], TOL["f32_out"][dtype]
    assert err(y, y_ref, dtype) <= st and err(dx, gr["dx"], dtype) <= st
    assert err(d1, gr["w1"]) <= f32 and err(da, gr["w2a"]) <= f32 and err(db, gr["w2b"]) <= f32


# ---- (d) the module takes this path --------------------------------------------------------------------------------------------
def test_module_low_rank_linear_runs_the_native_op(monkeypatch):
    from lycoris_amd import ops
    from lycoris_amd.modules import LokrModule
    torch.manual_seed(8)
    lin = nn.Linear(640, 1280, bias=False).to(DEV, torch.bfloat16)
    mod = LokrModule("t", lin, 1.0, 16, 8, factor=8).to(DEV)
    assert hasattr(mod, "lokr_w2_a") and not mod.use_w2
    with torch.no_grad():
        mod.lokr_w2_a.normal_(0, 0.3)
        mod.lokr_w2_b.normal_(0, 0.3)
    calls = []
    real = ops.lokr_linear_lr
    monkeypatch.setattr(ops, "lokr_linear_lr", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    x = torch.randn(64, 640, device=DEV, dtype=torch.bfloat16)
    y = mod(x)
    assert calls, "LokrModule did not take the low-rank native path"
    w = mod.get_diff_weight()[0].to(torch.float32)
    want = lin(x).float() + x.float() @ w.t()
    torch.cuda.synchronize()
    assert float((y.float() - want).norm() / want.norm()) <= 5e-3


# ---- (e) Conv2d: w2 = w2a [c, r] @ w2b [r, d*kh*kw] -----------------------------------------------------------------------------
def test_chain_group_with_window_taps():
    """Conv2d: dW2 arrives window-major [c, kh*kw, d], w2b and its gradient are [r, d*kh*kw] with column v * taps + tap"""
    from lycoris_amd import _native as N
    gen = torch.Generator().manual_seed(41)
    shapes = [(40, 40, 8, 9), (160, 160, 16, 9), (80, 40, 4, 9), (16, 24, 3, 4), (40, 120, 16, 1)]
    items = (N.LokrLrChainItem * len(shapes))()
    keep, want = [], []
    for k, (c, d, r, taps) in enumerate(shapes):
        dw2, dw264 = rnd((c, taps, d), torch.float32, gen)
        a, a64 = rnd((c, r), torch.float32, gen, 0.3)
        b, b64 = rnd((r, d * taps), torch.float32, gen, 0.3)
        da, db = torch.zeros(c, r, device=DEV), torch.zeros(r, d * taps, device=DEV)
        items[k] = N.LokrLrChainItem(N.ptr(dw2), N.ptr(a), N.ptr(b), N.ptr(da), N.ptr(db), c, d, r, taps)
        keep.append((dw2, a, b, da, db))
        g_ref = dw264.transpose(0, 2, 1).reshape(c, d * taps)  # [c, (v, tap)]: the reference's column order
        want.append((g_ref @ b64.T, a64.T @ g_ref))
    N.call("lyc_lokr_lr_chain_group", ctypes.cast(items, ctypes.c_void_p), len(shapes), N.stream_ptr(torch.device(DEV)))
    torch.cuda.synchronize()
    for k, ((_, _, _, da, db), (wa, wb)) in enumerate(zip(keep, want)):
        assert err(da, wa) <= 2e-6, (k, shapes[k], err(da, wa))
        assert err(db, wb) <= 2e-6, (k, shapes[k], err(db, wb))


# (B, C, H, O, k, stride, r): SDXL resnet convs at three resolutions, a downsample conv, a channel-changing up-block conv
CONV_LR = [(1, 320, 128, 320, 3, 1, 16), (1, 1280, 32, 1280, 3, 1, 16), (1, 640, 64, 640, 3, 2, 16), (1, 960, 64, 640, 3, 1, 16),
           (2, 64, 12, 128, 3, 1, 4)]


@DT16
@pytest.mark.parametrize("shape", CONV_LR, ids=[f"B{s[0]}_{s[1]}x{s[2]}to{s[3]}_k{s[4]}s{s[5]}_r{s[6]}" for s in CONV_LR])
def test_conv_op_matches_the_oracle(shape, dtype):
    from lycoris_amd import ops
    B, C, H, O, k, st, r = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x, x64 = rnd((B, C, H, H), dtype, gen)
    Ho = (H + 2 * (k // 2) - k) // st + 1
    g, g64 = rnd((B, O, Ho, Ho), dtype, gen, 1.0 / np.sqrt(O))
    w1, w1_64 = rnd((8, 8), torch.float32, gen, 0.3)
    w2a, a64 = rnd((O // 8, r), torch.float32, gen, 0.3)
    w2b, b64 = rnd((r, (C // 8) * k * k), torch.float32, gen, 0.1)
    x.requires_grad_(True)
    ps = [nn.Parameter(t) for t in (w1, w2a, w2b)]
    ext = ops._DISPATCH["ext"] if ops._cpp() else None
    native = ext.lokr_conv2d_lr_ok(x, ps[0], ps[1], ps[2], [k, k], [st, st], [k // 2, k // 2], [1, 1])
    # stride 2 is not a geometry of the patch kernels: ops.lokr_conv2d_lr then forms the product and runs the row-gather kernels
    assert native == (st == 1), "which path a layer takes is part of the contract"
    y = ops.lokr_conv2d_lr(x, ps[0], ps[1], ps[2], 1.0, (k, k), (st, st), (k // 2, k // 2), (1, 1))
    dx, d1, da, db = torch.autograd.grad(y, [x] + ps, g)
    torch.cuda.synchronize()
    ca = {"stride": st, "padding": k // 2, "dilation": 1}
    y_ref = oracle.lokr.forward(x64, w1=w1_64, w2a=a64, w2b=b64, scale=1.0, kshape=(k, k), conv_args=ca)
    gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2a=a64, w2b=b64, scale=1.0, kshape=(k, k), conv_args=ca)
    stol, f32 = TOL["store_out"][dtype], TOL["f32_out"][dtype]
    assert err(y, y_ref, dtype) <= stol, err(y, y_ref, dtype)
    assert err(dx, gr["dx"], dtype) <= stol, err(dx, gr["dx"], dtype)
    assert err(d1, gr["w1"]) <= f32, err(d1, gr["w1"])
    assert err(da, gr["w2a"]) <= f32, err(da, gr["w2a"])
    assert err(db, gr["w2b"]) <= f32, err(db, gr["w2b"])


def test_parked_low_rank_conv_layers_match_the_immediate_path():
    from lycoris_amd import ops
    torch.manual_seed(9)
    mk = lambda *s, sc: nn.Parameter(torch.randn(*s, device=DEV) * sc)
    layers = [(mk(8, 8, sc=0.3), mk(16, 4, sc=0.3), mk(4, 8 * 9, sc=0.2)) for _ in range(3)]
    # a full-matrix conv layer in the same park list (its w2 and .grad in channels_last memory: the layout the kernels accumulate in)
    full = (mk(8, 8, sc=0.3), nn.Parameter((torch.randn(16, 8, 3, 3, device=DEV) * 0.1).contiguous(memory_format=torch.channels_last)))
    params = [p for l in layers for p in l] + list(full)
    x = (torch.randn(2, 64, 12, 12, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    gy = (torch.randn(2, 128, 12, 12, device=DEV) * 0.1).to(torch.bfloat16)
    geo = ((1, 1), (1, 1), (1, 1))

    def net():
        ys = [ops.lokr_conv2d_lr(x, w1, a, b, 0.5, (3, 3), *geo) for w1, a, b in layers + [layers[1]]]
        ys.append(ops.lokr_conv2d(x, full[0], full[1], 0.5, *geo))
        return sum(ys)

    def run(defer):
        seen = []
        for p in params:
            p.grad = torch.zeros_like(p)
        x.grad = None
        ops.fused_grad_accumulation(True, callback=lambda p: seen.append(id(p)))
        ops.deferred_weight_gradients(defer)
        try:
            net().backward(gy)
            assert ops._DISPATCH["ext"].deferred_pending() == 0
            torch.cuda.synchronize()
            return x.grad.clone(), [p.grad.clone() for p in params], seen
        finally:
            ops.fused_grad_accumulation(False, None)
            ops.deferred_weight_gradients(True, 48)

    want = torch.autograd.grad(sum([ops.lokr_conv2d(x, w1, (a @ b).reshape(16, 8, 3, 3), 0.5, *geo) for w1, a, b in layers + [layers[1]]] +
                                   [ops.lokr_conv2d(x, full[0], full[1], 0.5, *geo)]), params, gy)
    dx0, g0, seen0 = run(False)
    dx1, g1, seen1 = run(True)
    assert torch.equal(dx0, dx1)
    for u, v, w in zip(g0, g1, want):
        assert float(u.abs().max()) > 0
        assert float((u - v).norm() / u.norm()) <= 2e-5
        assert float((v - w).norm() / w.norm()) <= 3e-4, float((v - w).norm() / w.norm())
    assert sorted(seen0) == sorted(seen1) == sorted(id(p) for p in params)


def test_module_low_rank_conv_runs_the_native_op(monkeypatch):
    from lycoris_amd import ops
    from lycoris_amd.modules import LokrModule
    torch.manual_seed(8)
    conv = nn.Conv2d(320, 640, 3, padding=1, bias=False).to(DEV, torch.bfloat16)
    mod = LokrModule("t", conv, 1.0, 16, 8, factor=8).to(DEV)
    assert hasattr(mod, "lokr_w2_a") and not mod.use_w2 and not mod.tucker
    with torch.no_grad():
        mod.lokr_w2_a.normal_(0, 0.3)
        mod.lokr_w2_b.normal_(0, 0.1)
    calls = []
    real = ops._OPS["lokr_conv2d_lr"] if ops._cpp() else None
    monkeypatch.setitem(ops._OPS, "lokr_conv2d_lr", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    x = torch.randn(1, 320, 32, 32, device=DEV, dtype=torch.bfloat16)
    y = mod(x)
    assert calls, "LokrModule did not take the low-rank Conv2d native path"
    w = mod.get_diff_weight()[0].to(torch.bfloat16)
    want = conv(x).float() + torch.nn.functional.conv2d(x, w, padding=1).float()
    torch.cuda.synchronize()
    assert float((y.float() - want).norm() / want.norm()) <= 1e-2


# ---- (f) decompose_both: w1 = w1a @ w1b as well (lokr_linear_lr2) ----------------------------------------------------------------
LR2_SHAPES = [(1024, 8, 160, 160, 16, 2), (77, 8, 160, 256, 16, 4), (333, 8, 40, 24, 4, 2), (64, 4, 32, 48, 8, 1)]  # (M, a, c, d, r2, r1)


@DT16
@pytest.mark.parametrize("shape", LR2_SHAPES, ids=[f"M{s[0]}_a{s[1]}_c{s[2]}_d{s[3]}_r{s[4]}_r1{s[5]}" for s in LR2_SHAPES])
def test_decompose_both_op_matches_the_oracle(shape, dtype):
    from lycoris_amd import ops
    M, a, c, d, r2, r1 = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x, x64 = rnd((M, a * d), dtype, gen)
    g, g64 = rnd((M, a * c), dtype, gen, 0.1)
    w1a, a164 = rnd((a, r1), torch.float32, gen, 0.5)
    w1b, b164 = rnd((r1, a), torch.float32, gen, 0.5)
    w2a, a264 = rnd((c, r2), torch.float32, gen, 0.3)
    w2b, b264 = rnd((r2, d), torch.float32, gen, 0.3)
    x.requires_grad_(True)
    ps = [nn.Parameter(t) for t in (w1a, w1b, w2a, w2b)]
    y = ops.lokr_linear_lr2(x, *ps, 0.7)
    grads = torch.autograd.grad(y, [x] + ps, g)
    torch.cuda.synchronize()
    kw = dict(w1a=a164, w1b=b164, w2a=a264, w2b=b264, scale=0.7)
    y_ref = oracle.lokr.forward(x64, **kw)
    gr = oracle.lokr.backward(x64, g64, **kw)
    st, f32 = TOL["store_out"][dtype], TOL["f32_out"][dtype]
    assert err(y, y_ref, dtype) <= st and err(grads[0], gr["dx"], dtype) <= st
    for got, key in zip(grads[1:], ("w1a", "w1b", "w2a", "w2b")):
        assert err(got, gr[key]) <= f32, (key, err(got, gr[key]))


def test_decompose_both_parked_layers_match_autograd_through_the_products():
    from lycoris_amd import ops
    torch.manual_seed(12)
    mk = lambda *s, sc: nn.Parameter(torch.randn(*s, device=DEV) * sc)
    layers = [(mk(8, 2, sc=0.5), mk(2, 8, sc=0.5), mk(16, 4, sc=0.3), mk(4, 16, sc=0.3)) for _ in range(3)]
    params = [p for l in layers for p in l]
    x = (torch.randn(96, 128, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    gy = (torch.randn(96, 128, device=DEV) * 0.1).to(torch.bfloat16)

    def net(materialise):
        h = x
        for l in layers + [layers[0]]:  # the first layer is applied twice
            if materialise:
                h = h + ops.lokr_linear(h, l[0] @ l[1], l[2] @ l[3], 0.5)
            else:
                h = h + ops.lokr_linear_lr2(h, *l, 0.5)
        return h

    want = torch.autograd.grad(net(True), params, gy)

    def run(defer):
        seen = []
        for p in params:
            p.grad = torch.zeros_like(p)
        x.grad = None
        ops.fused_grad_accumulation(True, callback=lambda p: seen.append(id(p)))
        ops.deferred_weight_gradients(defer)
        try:
            net(False).backward(gy)
            assert ops._DISPATCH["ext"].deferred_pending() == 0
            torch.cuda.synchronize()
            return x.grad.clone(), [p.grad.clone() for p in params], seen
        finally:
            ops.fused_grad_accumulation(False, None)
            ops.deferred_weight_gradients(True, 48)

    dx0, g0, seen0 = run(False)
    dx1, g1, seen1 = run(True)
    assert torch.equal(dx0, dx1)
    for u, v, w in zip(g0, g1, want):
        assert float(u.abs().max()) > 0
        assert float((u - v).norm() / u.norm()) <= 2e-5
        assert float((v - w).norm() / w.norm()) <= 3e-4, float((v - w).norm() / w.norm())
    assert sorted(seen0) == sorted(seen1) == sorted(id(p) for p in params)
