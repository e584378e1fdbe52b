"""GPU parity: LoHa Linear and the Conv2d forms of LoCon / LoHa / LoKr (ops level, all dtypes) vs the oracle."""
import numpy as np
import pytest
import torch

import oracle
from gpu_util import TOL, check, err, rnd, loha_cast_pair

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
IDS = ["f32", "bf16", "f16"]

LOHA_SHAPES = [(64, 64, 128, 8), (200, 320, 640, 32), (77, 200, 72, 5), (33, 50, 70, 40), (1, 128, 128, 4),
               (48, 2560, 2048, 16),   # 1280 tiles: 2 row tiles per workgroup in the factor-gradient kernel
               (40, 4160, 4096, 8),    # 4160 tiles: 4 x 2 tiles per workgroup, ragged last column group
               # gemm16d's 2-D tile -> XCD map (round 6): XCD grids 4 x 2, 8 x 1 and 2 x 4 over the output with RAGGED blocks (tile counts
               # that do not divide by the grid: surplus workgroups return) and ragged last row / column tiles, in all three contractions
               (1000, 320, 704, 8), (2200, 192, 128, 4), (520, 1344, 320, 8)]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("shape", LOHA_SHAPES, ids=[str(s) for s in LOHA_SHAPES])
def test_loha_linear(shape, dtype):
    from lycoris_amd import ops
    M, I, O, r = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
    w1a, a1 = rnd((O, r), torch.float32, gen, 0.1)
    w1b, b1 = rnd((r, I), torch.float32, gen, 1.0)
    w2a, a2 = rnd((O, r), torch.float32, gen, 0.1)
    w2b, b2 = rnd((r, I), torch.float32, gen, 1.0)
    alpha = 0.5
    ts = [x, w1a, w1b, w2a, w2b]
    for t in ts:
        t.requires_grad_(True)
    y = ops.loha_linear(x, w1a, w1b, w2a, w2b, alpha)
    grads = torch.autograd.grad(y, ts, g)
    torch.cuda.synchronize()
    y_ref = oracle.loha.forward(x64, a1, b1, a2, b2, alpha)
    ref = oracle.loha.backward(x64, g64, a1, b1, a2, b2, alpha)
    names = ["dx", "d_w1a", "d_w1b", "d_w2a", "d_w2b"]
    errs = {"y": err(y, y_ref, dtype), "dx": err(grads[0], ref[0], dtype)}
    bounds = {"y": TOL["loha_store"][dtype], "dx": TOL["loha_store"][dtype]}
    for n, gr, rf in zip(names[1:], grads[1:], ref[1:]):
        errs[n] = err(gr, rf)
        bounds[n] = TOL["f32_out"][dtype]
    loha_cast_pair(errs, bounds, dtype, y, grads[0], x64, g64, (a1, b1, a2, b2), alpha)
    check(f"loha_linear[{shape},{dtype}]", errs, bounds)


# (B, C, H, W, O, k, stride, pad, dil)
CONV_SHAPES = [
    (2, 16, 12, 12, 32, 3, 1, 1, 1),
    (1, 24, 9, 11, 16, 3, 2, 1, 1),
    (2, 16, 8, 8, 24, 1, 1, 0, 1),
    (1, 8, 10, 9, 8, 3, 1, 2, 2),
    (3, 32, 16, 16, 64, 3, 1, 1, 1),
]


# shapes that take the implicit-GEMM LoKr path (a == b in {4, 8}, c, d multiples of 8): (..., factor)
LOKR_IMPLICIT_SHAPES = [
    (2, 32, 12, 12, 64, 3, 1, 1, 1, 4),     # 3x3 same
    (1, 64, 9, 11, 32, 3, 2, 1, 1, 4),      # stride 2, odd spatial size
    (1, 32, 10, 9, 32, 3, 1, 2, 2, 4),      # dilation 2
    (2, 64, 8, 8, 128, 3, 1, 1, 1, 8),      # factor 8, d = 8
    (1, 32, 7, 7, 32, 5, 1, 2, 1, 4),       # 5x5 window
    (1, 128, 6, 5, 64, 3, 1, 0, 1, 8),      # no padding (valid conv)
    (3, 320, 9, 9, 320, 3, 1, 1, 1, 8),     # SDXL resnet channels at a small spatial size (d = 40: ragged k-steps)
    (1, 64, 5, 5, 64, 3, 2, 0, 1, 4),       # stride 2 without padding
]


def _ca(s, p, d):
    return {"stride": s, "padding": p, "dilation": d}


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("shape", CONV_SHAPES, ids=[str(s) for s in CONV_SHAPES])
def test_locon_conv2d(shape, dtype):
    from lycoris_amd import ops
    B, C, H, W, O, k, s, p, d = shape
    r = 4
    gen = torch.Generator().manual_seed(sum(shape))
    x, x64 = rnd((B, C, H, W), dtype, gen)
    down, d64 = rnd((r, C, k, k), torch.float32, gen, 0.1)
    up, u64 = rnd((O, r, 1, 1), torch.float32, gen, 0.1)
    y_ref = oracle.locon.forward(x64, d64, u64, 1.25, _ca(s, p, d))
    g, g64 = rnd(y_ref.shape, dtype, gen, 1.0 / np.sqrt(O))
    for t in (x, down, up):
        t.requires_grad_(True)
    y = ops.locon_conv2d(x, down, up, 1.25, (s, s), (p, p), (d, d))
    dx, dd, du = torch.autograd.grad(y, [x, down, up], g)
    torch.cuda.synchronize()
    dx_r, dd_r, du_r = oracle.locon.backward(x64, g64, d64, u64, 1.25, _ca(s, p, d))
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, dx_r, dtype), "d_down": err(dd, dd_r), "d_up": err(du, du_r)}
    bounds = {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype], "d_down": TOL["f32_out"][dtype], "d_up": TOL["f32_out"][dtype]}
    check(f"locon_conv2d[{shape},{dtype}]", errs, bounds)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("shape", CONV_SHAPES, ids=[str(s) for s in CONV_SHAPES])
def test_lokr_conv2d(shape, dtype):
    from lycoris_amd import ops
    B, C, H, W, O, k, s, p, d = shape
    a_, b_ = 4, 4
    c_, d_ = O // a_, C // b_
    gen = torch.Generator().manual_seed(sum(shape) + 1)
    x, x64 = rnd((B, C, H, W), dtype, gen)
    w1, w1_64 = rnd((a_, b_), torch.float32, gen, 0.3)
    w2, w2_64 = rnd((c_, d_, k, k), torch.float32, gen, 0.1)
    y_ref = oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=0.9, kshape=(k, k), conv_args=_ca(s, p, d))
    g, g64 = rnd(y_ref.shape, dtype, gen, 1.0 / np.sqrt(O))
    for t in (x, w1, w2):
        t.requires_grad_(True)
    y = ops.lokr_conv2d(x, w1, w2, 0.9, (s, s), (p, p), (d, d))
    dx, dw1, dw2 = torch.autograd.grad(y, [x, w1, w2], g)
    torch.cuda.synchronize()
    gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=0.9, kshape=(k, k), conv_args=_ca(s, p, d))
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, gr["dx"], dtype), "dw1": err(dw1, gr["w1"]), "dw2": err(dw2, gr["w2"])}
    bounds = {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype], "dw1": TOL["f32_out"][dtype], "dw2": TOL["f32_out"][dtype]}
    check(f"lokr_conv2d[{shape},{dtype}]", errs, bounds)


@pytest.mark.parametrize("layout", ["contiguous", "channels_last"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", LOKR_IMPLICIT_SHAPES, ids=[str(s) for s in LOKR_IMPLICIT_SHAPES])
def test_lokr_conv2d_implicit(shape, dtype, layout):
    """The im2col-free path: pixel-row gather inside the kernels, all taps summed in fp32, w2 in [c, kh, kw, d] order."""
    from lycoris_amd import ops
    B, C, H, W, O, k, s, p, d, f = shape
    c_, d_ = O // f, C // f
    gen = torch.Generator().manual_seed(sum(shape) + 7)
    x, x64 = rnd((B, C, H, W), dtype, gen)
    w1, w1_64 = rnd((f, f), torch.float32, gen, 0.3)
    w2, w2_64 = rnd((c_, d_, k, k), torch.float32, gen, 0.1)
    assert ops._lokr_conv_implicit_ok(x, w1, w2)
    if layout == "channels_last":
        x = x.contiguous(memory_format=torch.channels_last)
        w2 = w2.contiguous(memory_format=torch.channels_last)
    y_ref = oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=0.9, kshape=(k, k), conv_args=_ca(s, p, d))
    g, g64 = rnd(y_ref.shape, dtype, gen, 1.0 / np.sqrt(O))
    for t in (x, w1, w2):
        t.requires_grad_(True)
    y = ops.lokr_conv2d(x, w1, w2, 0.9, (s, s), (p, p), (d, d))
    dx, dw1, dw2 = torch.autograd.grad(y, [x, w1, w2], g)
    torch.cuda.synchronize()
    gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=0.9, kshape=(k, k), conv_args=_ca(s, p, d))
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, gr["dx"], dtype), "dw1": err(dw1, gr["w1"]), "dw2": err(dw2, gr["w2"])}
    bounds = {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype], "dw1": TOL["f32_out"][dtype], "dw2": TOL["f32_out"][dtype]}
    check(f"lokr_conv2d_implicit[{shape},{dtype},{layout}]", errs, bounds)


# shapes that take the implicit-GEMM LoCon path (C % 16 == 0, O % 8 == 0, rank in {4, 8, 12, 16}): (..., rank)
LOCON_IMPLICIT_SHAPES = [
    (2, 32, 12, 12, 64, 3, 1, 1, 1, 8),     # 3x3 same
    (1, 64, 9, 11, 32, 3, 2, 1, 1, 4),      # stride 2, odd spatial size
    (1, 32, 10, 9, 32, 3, 1, 2, 2, 8),      # dilation 2
    (2, 48, 8, 8, 128, 3, 1, 1, 1, 16),     # rank 16: K = 144 in the input-gradient kernel
    (1, 32, 7, 7, 40, 5, 1, 2, 1, 4),       # 5x5 window (25 taps), output width not a multiple of 16
    (1, 128, 6, 5, 64, 3, 1, 0, 1, 12),     # no padding, rank 12
    (3, 320, 9, 9, 320, 3, 1, 1, 1, 8),     # SDXL resnet channels at a small spatial size
    (1, 64, 5, 5, 64, 3, 2, 0, 1, 8),       # stride 2 without padding
    (1, 16, 40, 37, 24, 3, 1, 1, 1, 8),     # many pixels: several row slabs in the gradient kernel
]


@pytest.mark.parametrize("layout", ["contiguous", "channels_last"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", LOCON_IMPLICIT_SHAPES, ids=[str(s) for s in LOCON_IMPLICIT_SHAPES])
def test_locon_conv2d_implicit(shape, dtype, layout):
    """The im2col-free LoCon path: gathered reduce stage, LDS-gathered transposed convolution for dx, gathered factor
    gradient; lora_down in [r, kh, kw, C] order."""
    from lycoris_amd import ops
    B, C, H, W, O, k, s, p, d, r = shape
    gen = torch.Generator().manual_seed(sum(shape) + 11)
    x, x64 = rnd((B, C, H, W), dtype, gen)
    down, d64 = rnd((r, C, k, k), torch.float32, gen, 0.1)
    up, u64 = rnd((O, r, 1, 1), torch.float32, gen, 0.1)
    assert ops._locon_conv_implicit_ok(x, down, up)
    if layout == "channels_last":
        x = x.contiguous(memory_format=torch.channels_last)
        down = down.contiguous(memory_format=torch.channels_last)
    y_ref = oracle.locon.forward(x64, d64, u64, 1.25, _ca(s, p, d))
    g, g64 = rnd(y_ref.shape, dtype, gen, 1.0 / np.sqrt(O))
    for t in (x, down, up):
        t.requires_grad_(True)
    y = ops.locon_conv2d(x, down, up, 1.25, (s, s), (p, p), (d, d))
    dx, dd, du = torch.autograd.grad(y, [x, down, up], g)
    torch.cuda.synchronize()
    dx_r, dd_r, du_r = oracle.locon.backward(x64, g64, d64, u64, 1.25, _ca(s, p, d))
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, dx_r, dtype), "d_down": err(dd, dd_r), "d_up": err(du, du_r)}
    bounds = {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype], "d_down": TOL["f32_out"][dtype], "d_up": TOL["f32_out"][dtype]}
    check(f"locon_conv2d_implicit[{shape},{dtype},{layout}]", errs, bounds)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("shape", CONV_SHAPES[:3], ids=[str(s) for s in CONV_SHAPES[:3]])
def test_loha_conv2d(shape, dtype):
    from lycoris_amd import ops
    B, C, H, W, O, k, s, p, d = shape
    r = 4
    gen = torch.Generator().manual_seed(sum(shape) + 2)
    x, x64 = rnd((B, C, H, W), dtype, gen)
    w1a, a1 = rnd((O, r), torch.float32, gen, 0.1)
    w1b, b1 = rnd((r, C * k * k), torch.float32, gen, 1.0)
    w2a, a2 = rnd((O, r), torch.float32, gen, 0.1)
    w2b, b2 = rnd((r, C * k * k), torch.float32, gen, 1.0)
    wshape = (O, C, k, k)
    y_ref = oracle.loha.forward(x64, a1, b1, a2, b2, 0.5, wshape, _ca(s, p, d))
    g, g64 = rnd(y_ref.shape, dtype, gen, 1.0 / np.sqrt(O))
    ts = [x, w1a, w1b, w2a, w2b]
    for t in ts:
        t.requires_grad_(True)
    y = ops.loha_conv2d(x, w1a, w1b, w2a, w2b, 0.5, wshape, (s, s), (p, p), (d, d))
    grads = torch.autograd.grad(y, ts, g)
    torch.cuda.synchronize()
    ref = oracle.loha.backward(x64, g64, a1, b1, a2, b2, 0.5, wshape, _ca(s, p, d))
    errs = {"y": err(y, y_ref, dtype), "dx": err(grads[0], ref[0], dtype)}
    bounds = {"y": TOL["loha_store"][dtype], "dx": TOL["loha_store"][dtype]}
    for n, gr, rf in zip(["d_w1a", "d_w1b", "d_w2a", "d_w2b"], grads[1:], ref[1:]):
        errs[n] = err(gr, rf)
        bounds[n] = TOL["f32_out"][dtype]
    loha_cast_pair(errs, bounds, dtype, y, grads[0], x64, g64, (a1, b1, a2, b2), 0.5, wshape, _ca(s, p, d))
    check(f"loha_conv2d[{shape},{dtype}]", errs, bounds)


@pytest.mark.parametrize("layout", ["contiguous", "channels_last"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
@pytest.mark.parametrize("shape", [(2, 64, 9, 7, 128, 8), (1, 320, 16, 16, 640, 8), (3, 48, 5, 5, 32, 4)], ids=str)
def test_lokr_conv2d_pointwise_is_the_linear_op(shape, dtype, layout):
    """A 1x1 LoKr convolution runs the nn.Linear op on the NHWC pixel rows with the 4-D LEAF w2 [c, d, 1, 1] itself (round 6): packed
    operand planes, and -- in the training configuration -- the factor gradients accumulated straight into .grad by the grouped launch.
    Reference: lycoris/functional/lokr.py:154-247 (conv branch), modules/lokr.py:543-566."""
    from lycoris_amd import ops
    B, C, H, W, O, f = shape
    gen = torch.Generator().manual_seed(sum(shape) + 11)
    x, x64 = rnd((B, C, H, W), dtype, gen)
    w1, w1_64 = rnd((f, f), torch.float32, gen, 0.3)
    w2, w2_64 = rnd((O // f, C // f, 1, 1), torch.float32, gen, 0.1)
    if layout == "channels_last":
        x = x.contiguous(memory_format=torch.channels_last)
    ca = _ca(1, 0, 1)
    y_ref = oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=0.9, kshape=(1, 1), conv_args=ca)
    g, g64 = rnd(y_ref.shape, dtype, gen, 1.0 / np.sqrt(O))
    if layout == "channels_last":
        g = g.contiguous(memory_format=torch.channels_last)
    gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=0.9, kshape=(1, 1), conv_args=ca)
    bounds = {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype], "dw1": TOL["f32_out"][dtype], "dw2": TOL["f32_out"][dtype]}
    # plain autograd
    xs, w1s, w2s = x.clone().requires_grad_(True), w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    y = ops.lokr_conv2d(xs, w1s, w2s, 0.9, (1, 1), (0, 0), (1, 1))
    assert y.shape == (B, O, H, W)
    dx, dw1, dw2 = torch.autograd.grad(y, [xs, w1s, w2s], g)
    torch.cuda.synchronize()
    assert dw2.shape == w2.shape and dx.shape == x.shape
    check(f"lokr_pointwise[{shape},{dtype},{layout}]",
          {"y": err(y, y_ref, dtype), "dx": err(dx, gr["dx"], dtype), "dw1": err(dw1, gr["w1"]), "dw2": err(dw2, gr["w2"])}, bounds)
    # training configuration: leaves with fp32 .grad buffers, fused accumulation, deferred + grouped weight gradients
    w1p, w2p = torch.nn.Parameter(w1.clone()), torch.nn.Parameter(w2.clone())
    w1p.grad, w2p.grad = torch.zeros_like(w1p), torch.zeros_like(w2p)
    xe = x.clone().requires_grad_(True)
    ops.fused_grad_accumulation(True, None)
    try:
        ops.lokr_conv2d(xe, w1p, w2p, 0.9, (1, 1), (0, 0), (1, 1)).backward(g)
        torch.cuda.synchronize()
    finally:
        ops.fused_grad_accumulation(False, None)
    check(f"lokr_pointwise_fused[{shape},{dtype},{layout}]",
          {"dx": err(xe.grad, gr["dx"], dtype), "dw1": err(w1p.grad, gr["w1"]), "dw2": err(w2p.grad, gr["w2"])},
          {k: bounds[k] for k in ("dx", "dw1", "dw2")})


@pytest.mark.parametrize("layout", ["contiguous", "channels_last"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
@pytest.mark.parametrize("shape", [(2, 64, 9, 7, 128, 8), (1, 320, 16, 16, 640, 16), (3, 48, 5, 5, 32, 4)], ids=str)
def test_locon_conv2d_pointwise_is_the_linear_op(shape, dtype, layout):
    """A 1x1 LoCon convolution on a channels_last tensor runs the nn.Linear op on the NHWC pixel rows with the 4-D LEAVES lora_down
    [r, C, 1, 1] / lora_up [O, r, 1, 1] (round 6) -- fused accumulation into .grad and the grouped factor-gradient launch included.
    Reference: lycoris/functional/locon.py:64-85, modules/locon.py:286-332."""
    from lycoris_amd import ops
    B, C, H, W, O, r = shape
    gen = torch.Generator().manual_seed(sum(shape) + 13)
    x, x64 = rnd((B, C, H, W), dtype, gen)
    down, d64 = rnd((r, C, 1, 1), torch.float32, gen, 0.1)
    up, u64 = rnd((O, r, 1, 1), torch.float32, gen, 0.1)
    if layout == "channels_last":
        x = x.contiguous(memory_format=torch.channels_last)
    ca = _ca(1, 0, 1)
    y_ref = oracle.locon.forward(x64, d64, u64, 1.25, ca)
    g, g64 = rnd(y_ref.shape, dtype, gen, 1.0 / np.sqrt(O))
    if layout == "channels_last":
        g = g.contiguous(memory_format=torch.channels_last)
    dx_r, dd_r, du_r = oracle.locon.backward(x64, g64, d64, u64, 1.25, ca)
    bounds = {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype], "d_down": TOL["f32_out"][dtype], "d_up": TOL["f32_out"][dtype]}
    xs, ds, us = x.clone().requires_grad_(True), down.clone().requires_grad_(True), up.clone().requires_grad_(True)
    y = ops.locon_conv2d(xs, ds, us, 1.25, (1, 1), (0, 0), (1, 1))
    assert y.shape == (B, O, H, W)
    dx, dd, du = torch.autograd.grad(y, [xs, ds, us], g)
    torch.cuda.synchronize()
    assert dd.shape == down.shape and du.shape == up.shape and dx.shape == x.shape
    check(f"locon_pointwise[{shape},{dtype},{layout}]",
          {"y": err(y, y_ref, dtype), "dx": err(dx, dx_r, dtype), "d_down": err(dd, dd_r), "d_up": err(du, du_r)}, bounds)
    dp, upp = torch.nn.Parameter(down.clone()), torch.nn.Parameter(up.clone())
    dp.grad, upp.grad = torch.zeros_like(dp), torch.zeros_like(upp)
    xe = x.clone().requires_grad_(True)
    ops.fused_grad_accumulation(True, None)
    try:
        ops.locon_conv2d(xe, dp, upp, 1.25, (1, 1), (0, 0), (1, 1)).backward(g)
        torch.cuda.synchronize()
    finally:
        ops.fused_grad_accumulation(False, None)
    check(f"locon_pointwise_fused[{shape},{dtype},{layout}]",
          {"dx": err(xe.grad, dx_r, dtype), "d_down": err(dp.grad, dd_r), "d_up": err(upp.grad, du_r)},
          {k: bounds[k] for k in ("dx", "d_down", "d_up")})
