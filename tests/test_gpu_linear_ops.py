"""GPU parity: LoKr / LoCon / (IA)^3 Linear ops through the C ABI vs the numpy oracle."""
import numpy as np
import pytest
import torch

import oracle
from gpu_util import TOL, check, dev, err, rnd

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16]

# (M, a, b, c, d): out = a*c, in = b*d
LOKR_SHAPES = [
    (64, 8, 8, 32, 32),       # aligned, TQ=32 path
    (96, 8, 8, 160, 160),     # SDXL attn projection factors (1280 -> 1280)
    (40, 8, 8, 192, 40),      # SDXL 320-wide input, wide output tile (N % 64 == 0)
    (33, 4, 8, 20, 24),       # ragged rows, N not multiple of 16
    (7, 5, 3, 7, 9),          # nothing aligned: scalar load/store paths
    (130, 16, 2, 12, 50),     # a != b, TM = 64
    (19, 2, 32, 36, 5),       # large Gin
    (1, 8, 8, 40, 160),       # single row (time-embedding projections)
    # fast path (Gin == Gout | 16, K % 8 == 0) corner cases
    (50, 4, 4, 24, 200),      # G=4, two K chunks (160 + 40), N < 64
    (37, 16, 16, 20, 96),     # G=16, ragged rows, N not a multiple of 8
    (100, 8, 8, 72, 640),     # four full K chunks, N = 64 + 8
    (9, 2, 2, 100, 8),        # G=2, K = 8
    (20, 1, 1, 64, 64),       # G=1 (plain x @ w2^T scaled by w1[0,0])
    (64, 8, 8, 17, 328),      # K = 2 chunks + 8, odd N (scalar stores)
    (300, 8, 8, 1280, 160),   # ff.net.0.proj factors at reduced M
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", LOKR_SHAPES, ids=[str(s) for s in LOKR_SHAPES])
def test_lokr_linear(shape, dtype):
    from lycoris_amd import ops
    M, a, b, c, d = shape
    gen = torch.Generator().manual_seed(hash(shape) % 10000)
    x, x64 = rnd((M, b * d), dtype, gen)
    g, g64 = rnd((M, a * c), dtype, gen, 1.0 / np.sqrt(a * c))
    w1, w1_64 = rnd((a, b), torch.float32, gen, 0.3)
    w2, w2_64 = rnd((c, d), torch.float32, gen, 0.1)
    alpha = 0.75
    x.requires_grad_(True); w1.requires_grad_(True); w2.requires_grad_(True)
    y = ops.lokr_linear(x, w1, w2, alpha)
    dx, dw1, dw2 = torch.autograd.grad(y, [x, w1, w2], g)
    torch.cuda.synchronize()
    y_ref = oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=alpha)
    gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=alpha)
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, gr["dx"], dtype), "dw1": err(dw1, gr["w1"]), "dw2": err(dw2, gr["w2"])}
    bounds = {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype], "dw1": TOL["f32_out"][dtype], "dw2": TOL["f32_out"][dtype]}
    check(f"lokr_linear[{shape},{dtype}]", errs, bounds)


LOCON_SHAPES = [  # (M, I, O, r)
    (64, 64, 128, 16),
    (200, 320, 640, 16),
    (77, 2048, 1280, 4),
    (33, 50, 70, 3),         # unaligned
    (130, 96, 200, 40),      # rank > 32
    (5, 1280, 320, 136),     # rank > 128 (two skinny tiles)
    (1, 1280, 1280, 8),
    # fused rank-r path (16-bit, I % 8 == 0 forward / O % 8 == 0 backward, r <= 64) corner cases
    (300, 1280, 1280, 16),   # SDXL attention projection, ragged last row tile
    (100, 72, 40, 6),        # rank not a multiple of 4: element-wise factor loads
    (50, 24, 36, 12),        # forward fused, backward on the general kernels (O % 8 != 0)
    (33, 40, 50, 8),         # output width not a multiple of 4: scalar stores
    (70, 40, 24, 20),        # two rank tiles
    (45, 128, 72, 64),       # four rank tiles
    (9000, 64, 32, 8),       # 32-row workgroups (M >= 8192), ragged end, many row slabs in the gradient kernel
    (4100, 64, 64, 16),      # 4-wave workgroups
    (260, 2560, 640, 32),    # long K, rank 32
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", LOCON_SHAPES, ids=[str(s) for s in LOCON_SHAPES])
def test_locon_linear(shape, dtype):
    from lycoris_amd import ops
    M, I, O, r = shape
    gen = torch.Generator().manual_seed(hash(shape) % 10000)
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
    down, d64 = rnd((r, I), torch.float32, gen, 0.05)
    up, u64 = rnd((O, r), torch.float32, gen, 0.05)
    alpha = 1.5
    x.requires_grad_(True); down.requires_grad_(True); up.requires_grad_(True)
    y = ops.locon_linear(x, down, up, alpha)
    dx, dd, du = torch.autograd.grad(y, [x, down, up], g)
    torch.cuda.synchronize()
    y_ref = oracle.locon.forward(x64, d64, u64, alpha)
    dx_ref, dd_ref, du_ref = oracle.locon.backward(x64, g64, d64, u64, alpha)
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, dx_ref, dtype), "d_down": err(dd, dd_ref), "d_up": err(du, du_ref)}
    bounds = {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype], "d_down": TOL["f32_out"][dtype], "d_up": TOL["f32_out"][dtype]}
    check(f"locon_linear[{shape},{dtype}]", errs, bounds)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_locon_linear_no_input_grad(dtype):
    """first layer of a network: x does not require grad -> the backward call gets dx = NULL"""
    from lycoris_amd import ops
    M, I, O, r = 150, 320, 640, 16
    gen = torch.Generator().manual_seed(5)
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
    down, d64 = rnd((r, I), torch.float32, gen, 0.05)
    up, u64 = rnd((O, r), torch.float32, gen, 0.05)
    down.requires_grad_(True); up.requires_grad_(True)
    y = ops.locon_linear(x, down, up, 0.5)
    dd, du = torch.autograd.grad(y, [down, up], g)
    torch.cuda.synchronize()
    _, dd_ref, du_ref = oracle.locon.backward(x64, g64, d64, u64, 0.5)
    check("locon_linear[no dx]", {"d_down": err(dd, dd_ref), "d_up": err(du, du_ref)},
          {"d_down": TOL["f32_out"][dtype], "d_up": TOL["f32_out"][dtype]})


CHAN_SHAPES = [((64, 128), -1), ((3, 77, 1280), -1), ((33, 50), -1), ((2, 16, 9, 11), 1), ((4, 320, 16, 16), 1)]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", CHAN_SHAPES, ids=[str(s[0]) for s in CHAN_SHAPES])
@pytest.mark.parametrize("with_bias", [False, True])
def test_chan_affine(case, dtype, with_bias):
    from lycoris_amd import ops
    shape, cd = case
    gen = torch.Generator().manual_seed(sum(shape))
    C = shape[cd]
    a, a64 = rnd(shape, dtype, gen)
    g, g64 = rnd(shape, dtype, gen)
    w, w64 = rnd((C,), torch.float32, gen, 0.3)
    bias, b64 = rnd((C,), torch.float32, gen, 0.5) if with_bias else (None, None)
    s0, mult = (1.0, 0.7) if with_bias else (0.0, 1.3)
    a.requires_grad_(True); w.requires_grad_(True)
    out = ops.chan_affine(a, w, bias, s0, mult, cd)
    da, dw = torch.autograd.grad(out, [a, w], g)
    torch.cuda.synchronize()
    bs = [1] * len(shape); bs[cd] = C
    wm = (w64 * mult).reshape(bs)
    bb = b64.reshape(bs) if with_bias else 0.0
    out_ref = a64 * (s0 + wm) - bb * wm
    da_ref = g64 * (s0 + wm)
    axes = tuple(i for i in range(len(shape)) if i != (cd % len(shape)))
    dw_ref = (g64 * (a64 - bb)).sum(axis=axes) * mult
    errs = {"out": err(out, out_ref, dtype), "da": err(da, da_ref, dtype), "dw": err(dw, dw_ref)}
    bounds = {"out": TOL["store_out"][dtype], "da": TOL["store_out"][dtype], "dw": TOL["f32_out"][dtype]}
    check(f"chan_affine[{shape},{dtype},{with_bias}]", errs, bounds)


BWD_SHAPES = [(1024, 1280), (1024, 5120), (77, 1280), (333, 640), (1, 1280), (19, 72)]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", BWD_SHAPES, ids=[f"{m}x{c}" for m, c in BWD_SHAPES])
@pytest.mark.parametrize("half", ["both", "da_only", "dw_only"])
def test_chan_bwd_one_pass_through_the_c_abi(shape, dtype, half):
    """lyc_chan_bwd (round 6): da = g * (s0 + w mult) and dw += mult * sum g * (a - bias) in one pass over g, either half optional;
    SDXL (IA)^3 layer shapes at full size (to_k / to_v 1024 x 1280 and 77 x 1280, ff.net.2's input 1024 x 5120) and ragged rows"""
    from lycoris_amd import _native as N
    M, C = shape
    gen = torch.Generator().manual_seed(M + C)
    a, a64 = rnd((M, C), dtype, gen)
    g, g64 = rnd((M, C), dtype, gen, 0.2)
    w, w64 = rnd((C,), torch.float32, gen, 0.3)
    bias, b64 = rnd((C,), torch.float32, gen, 0.5)
    s0, mult = 1.0, 0.7
    da = torch.full((M, C), float("nan"), dtype=dtype, device=a.device) if half != "dw_only" else None
    dw0, dw0_64 = rnd((C,), torch.float32, gen)  # the kernel ADDS into dw
    dw = dw0.clone() if half != "da_only" else None
    N.call("lyc_chan_bwd", N.ptr(g), N.ptr(a), N.ptr(w), N.ptr(bias), N.ptr(da), N.ptr(dw), M, C, 1, s0, mult, N.dtype_code(dtype),
           N.stream_ptr(a.device))
    torch.cuda.synchronize()
    errs, bounds = {}, {}
    if da is not None:
        errs["da"], bounds["da"] = err(da, g64 * (s0 + w64 * mult), dtype), TOL["store_out"][dtype]
    if dw is not None:
        errs["dw"], bounds["dw"] = err(dw, dw0_64 + (g64 * (a64 - b64)).sum(axis=0) * mult), TOL["f32_out"][dtype]
    check(f"chan_bwd[{shape},{dtype},{half}]", errs, bounds)
