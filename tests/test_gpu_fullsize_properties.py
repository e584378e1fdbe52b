"""GPU parity at BASELINE.json's FULL sizes through size-independent properties (the float64 oracle takes minutes at
these sizes, so it checks the small cases; here the kernels are checked against identities that hold at any size).

For every adapter delta = f(x; factors), f is linear in x and linear in each factor separately, so with an upstream
gradient g and dx, d_factor = the op's backward:

    <g, y> = <dx, x> = <d_w1, w1> = <d_w2, w2> (= <d_down, down> = <d_up, up> for LoCon)          (adjoint / Euler)
    f(x1 + x2) = f(x1) + f(x2)                                                                     (linearity)
    LoKr with w1 = I and w2 = I (square layer) is alpha * x; LoCon with down = [I_r 0], up = [I_r; 0] copies r columns

The inner products are evaluated in float64 from the bf16 / fp32 tensors the kernels produced.  The upstream gradient is
g = y itself, so that <g, y> = |y|^2 is a large positive sum (with a random g the inner product is a zero-mean random
walk of the same size as its own rounding noise and a relative bound would be meaningless).  Bounds: 16-bit stored
y / dx carry one rounding each (2^-9 relative per element, random sign -> ~1e-3 on these sums), factor gradients are
fp32."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# SDXL UNet 1024x1024 bs=1 layer shapes (benchmarks/sdxl_shapes.py): (M, I, O)
LINEAR = [(1024, 1280, 1280), (1024, 1280, 10240), (1024, 5120, 1280), (4096, 640, 5120), (4096, 2560, 640), (77, 2048, 1280)]
# (C, H, O, k, stride)
CONV = [(320, 128, 320, 3, 1), (1280, 32, 1280, 3, 1), (640, 64, 640, 3, 2), (1920, 32, 1280, 3, 1)]


def dot(a, b):
    return float((a.detach().double() * b.detach().double()).sum())


def close(a, b, tol):
    assert abs(a - b) <= tol * max(abs(a), abs(b), 1e-30), (a, b, tol)


@pytest.mark.parametrize("shape", LINEAR, ids=[str(s) for s in LINEAR])
def test_lokr_linear_fullsize_adjoint_and_linearity(shape):
    from lycoris_amd import ops
    M, I, O = shape
    gen = torch.Generator(device=DEV).manual_seed(M + I + O)
    x = torch.randn(M, I, device=DEV, dtype=torch.bfloat16, generator=gen).requires_grad_(True)
    x2 = torch.randn(M, I, device=DEV, dtype=torch.bfloat16, generator=gen)
    w1 = (torch.randn(8, 8, device=DEV, generator=gen) * 0.3).requires_grad_(True)
    w2 = (torch.randn(O // 8, I // 8, device=DEV, generator=gen) * 0.05).requires_grad_(True)
    y = ops.lokr_linear(x, w1, w2, 0.75)
    g = y.detach().clone()
    dx, dw1, dw2 = torch.autograd.grad(y, [x, w1, w2], g)
    gy = dot(g, y)
    # y and dx are rounded to bf16 per element (independent roundings average out in the inner product): 2e-3
    close(gy, dot(dx, x), 2e-3)
    # the factor gradients are fp32 and use the exact y = hi+lo arithmetic: they agree with each other far tighter
    close(dot(dw1, w1), dot(dw2, w2), 2e-5)
    close(gy, dot(dw2, w2), 2e-3)
    # linearity in x (both sides rounded to bf16 once per term)
    ya = ops.lokr_linear(x.detach() + x2, w1.detach(), w2.detach(), 0.75).double()
    yb = y.detach().double() + ops.lokr_linear(x2, w1.detach(), w2.detach(), 0.75).double()
    rel = float((ya - yb).norm() / yb.norm())
    assert rel < 8e-3, rel  # x + x2 is itself rounded to bf16 before the call
    if I == O:  # identity factors: the adapter is alpha * x exactly (alpha = 0.5 is a power of two)
        eye1, eye2 = torch.eye(8, device=DEV), torch.eye(I // 8, device=DEV)
        yi = ops.lokr_linear(x.detach(), eye1, eye2, 0.5)
        assert torch.equal(yi, (x.detach().float() * 0.5).to(torch.bfloat16))


@pytest.mark.parametrize("shape", CONV, ids=[str(s) for s in CONV])
def test_lokr_conv2d_fullsize_adjoint(shape):
    from lycoris_amd import ops
    C, H, O, k, s = shape
    gen = torch.Generator(device=DEV).manual_seed(C + H + O)
    x = torch.randn(1, C, H, H, device=DEV, dtype=torch.bfloat16, generator=gen).requires_grad_(True)
    w1 = (torch.randn(8, 8, device=DEV, generator=gen) * 0.3).requires_grad_(True)
    w2 = (torch.randn(O // 8, C // 8, k, k, device=DEV, generator=gen) * 0.05)
    w2 = w2.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert ops._lokr_conv_implicit_ok(x, w1, w2)
    y = ops.lokr_conv2d(x, w1, w2, 1.0, (s, s), (k // 2, k // 2), (1, 1))
    g = y.detach().clone()
    dx, dw1, dw2 = torch.autograd.grad(y, [x, w1, w2], g)
    gy = dot(g, y)
    close(gy, dot(dx, x), 2e-3)
    close(dot(dw1, w1), dot(dw2, w2), 2e-5)
    close(gy, dot(dw1, w1), 2e-3)
    # against the im2col lowering (an independent code path through the same C ABI) on the same inputs
    xr = x.detach().clone().requires_grad_(True)
    w1r, w2r = w1.detach().clone().requires_grad_(True), w2.detach().contiguous().clone().requires_grad_(True)
    yr = ops._AdapterConv2d.apply(ops._LokrCore, 1.0, ops._geom((k, k), (s, s), (k // 2, k // 2), (1, 1)), xr, w1r,
                                  w2r.reshape(w2r.shape[0], -1))
    dxr, dw1r, dw2r = torch.autograd.grad(yr, [xr, w1r, w2r], g)
    for got, ref, tol in ((y, yr, 2e-3), (dx, dxr, 2e-3), (dw1, dw1r, 1e-4), (dw2, dw2r, 1e-4)):
        rel = float((got.detach().double() - ref.detach().double()).norm() / ref.detach().double().norm())
        assert rel < tol, rel


@pytest.mark.parametrize("shape", LINEAR[:3], ids=[str(s) for s in LINEAR[:3]])
def test_locon_linear_fullsize_adjoint(shape):
    from lycoris_amd import ops
    M, I, O = shape
    r = 16
    gen = torch.Generator(device=DEV).manual_seed(M + I + O + 1)
    x = torch.randn(M, I, device=DEV, dtype=torch.bfloat16, generator=gen).requires_grad_(True)
    down = (torch.randn(r, I, device=DEV, generator=gen) * 0.05).requires_grad_(True)
    up = (torch.randn(O, r, device=DEV, generator=gen) * 0.05).requires_grad_(True)
    y = ops.locon_linear(x, down, up, 0.5)
    g = y.detach().clone()
    dx, dd, du = torch.autograd.grad(y, [x, down, up], g)
    gy = dot(g, y)
    close(gy, dot(dx, x), 2e-3)
    close(dot(dd, down), dot(du, up), 2e-5)
    close(gy, dot(du, up), 2e-3)
    # selector factors copy the first r input columns to the first r output columns
    sel_d = torch.zeros(r, I, device=DEV)
    sel_d[:, :r] = torch.eye(r, device=DEV)
    sel_u = torch.zeros(O, r, device=DEV)
    sel_u[:r] = torch.eye(r, device=DEV)
    ys = ops.locon_linear(x.detach(), sel_d, sel_u, 1.0)
    assert torch.equal(ys[:, :r], x.detach()[:, :r]) and not ys[:, r:].any()
