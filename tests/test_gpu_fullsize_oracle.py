"""GPU parity at the FULL BASELINE sizes against the float64 oracle.

The host picks tiles / row slabs / split-K factors from M, so M = 1024 / 4096 / 16384 take launch configurations the
small parity cases never see.  Every distinct nn.Linear shape of the SDXL workload (benchmarks/sdxl_shapes.py) and of
the SD1.5 workload (benchmarks/sd15_shapes.py: M = 16384 @ 320, 4096 @ 640, ...) is compared with the numpy oracle
(BLAS float64: seconds per case), for LoKr factor 8, LoCon dim 16 and LoHa dim 32 -- the BASELINE configurations.
Conv2d: full-size layers of both workloads (3x3 stride 1 / stride 2, 1x1 shortcut, SD1.5's 1x1 proj_in) against the
oracle, which is itself cross-checked once against torch's float64 CPU conv2d of the oracle's dW.
"""
import numpy as np
import pytest
import torch

import oracle
from benchmarks.sd15_shapes import distinct_linear_shapes
from benchmarks.sdxl_shapes import sdxl_unet_layers
from gpu_util import TOL, check, dev, err, rnd

pytestmark = pytest.mark.gpu


def _linear_shapes():
    seen, out = set(), []
    for l in sdxl_unet_layers(1):
        if l["kind"] == "linear":
            seen.add((l["M"], l["I"], l["O"]))
    for key in distinct_linear_shapes(4):
        seen.add(key)
    out = sorted(seen)
    return out


LINEAR = _linear_shapes()
IDS = [f"M{m}_{i}to{o}" for m, i, o in LINEAR]


def _bounds(dtype, names_store, names_f32, store="store_out"):
    b = {n: TOL[store][dtype] for n in names_store}
    b.update({n: TOL["f32_out"][dtype] for n in names_f32})
    return b


DT16 = pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])


@DT16
@pytest.mark.parametrize("shape", LINEAR, ids=IDS)
def test_lokr_linear_fullsize(shape, dtype):
    from lycoris_amd import ops
    M, I, O = shape
    a = b = 8
    c, d = O // 8, I // 8
    gen = torch.Generator().manual_seed(M + I + O)
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
    w1, w1_64 = rnd((a, b), torch.float32, gen, 0.3)
    w2, w2_64 = rnd((c, d), torch.float32, gen, 0.05)
    for t in (x, w1, w2):
        t.requires_grad_(True)
    y = ops.lokr_linear(x, w1, w2, 1.0)
    dx, dw1, dw2 = torch.autograd.grad(y, [x, w1, w2], g)
    torch.cuda.synchronize()
    y_ref = oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=1.0)
    gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=1.0)
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, gr["dx"], dtype), "dw1": err(dw1, gr["w1"]), "dw2": err(dw2, gr["w2"])}
    check(f"lokr_linear_full[{shape},{dtype}]", errs, _bounds(dtype, ["y", "dx"], ["dw1", "dw2"]))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", LINEAR, ids=IDS)
def test_locon_linear_fullsize(shape, dtype):
    from lycoris_amd import ops
    M, I, O = shape
    r = 16
    gen = torch.Generator().manual_seed(M + I + O + 1)
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
    down, d64 = rnd((r, I), torch.float32, gen, 0.05)
    up, u64 = rnd((O, r), torch.float32, gen, 0.05)
    for t in (x, down, up):
        t.requires_grad_(True)
    y = ops.locon_linear(x, down, up, 0.5)
    dx, dd, du = torch.autograd.grad(y, [x, down, up], g)
    torch.cuda.synchronize()
    y_ref = oracle.locon.forward(x64, d64, u64, 0.5)
    dx_ref, dd_ref, du_ref = oracle.locon.backward(x64, g64, d64, u64, 0.5)
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, dx_ref, dtype), "d_down": err(dd, dd_ref), "d_up": err(du, du_ref)}
    check(f"locon_linear_full[{shape},{dtype}]", errs, _bounds(dtype, ["y", "dx"], ["d_down", "d_up"]))


@DT16
@pytest.mark.parametrize("shape", LINEAR, ids=IDS)
def test_loha_linear_fullsize(shape, dtype):
    from lycoris_amd import ops
    M, I, O = shape
    r = 32
    gen = torch.Generator().manual_seed(M + I + O + 2)
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
    w1a, a1 = rnd((O, r), torch.float32, gen, 0.1)
    w1b, b1 = rnd((r, I), torch.float32, gen, 1.0)
    w2a, a2 = rnd((O, r), torch.float32, gen, 0.1)
    w2b, b2 = rnd((r, I), torch.float32, gen, 1.0)
    ts = [x, w1a, w1b, w2a, w2b]
    for t in ts:
        t.requires_grad_(True)
    y = ops.loha_linear(x, w1a, w1b, w2a, w2b, 0.25)
    grads = torch.autograd.grad(y, ts, g)
    torch.cuda.synchronize()
    y_ref = oracle.loha.forward(x64, a1, b1, a2, b2, 0.25)
    ref = oracle.loha.backward(x64, g64, a1, b1, a2, b2, 0.25)
    names = ["dx", "d_w1a", "d_w1b", "d_w2a", "d_w2b"]
    errs = {"y": err(y, y_ref, dtype), "dx": err(grads[0], ref[0], dtype)}
    for n, gr, rf in zip(names[1:], grads[1:], ref[1:]):
        errs[n] = err(gr, rf)
    bounds = _bounds(dtype, ["y", "dx"], names[1:], "loha_store")
    # the same quantities against the oracle evaluated WITH the reference's cast `get_weight(...).to(base_weight.dtype)`
    # (modules/loha.py:310, oracle.loha round_dw): north-star bound 1e-3
    y_cast = oracle.loha.forward(x64, a1, b1, a2, b2, 0.25, round_dw=str(dtype))
    dx_cast = oracle.loha.backward(x64, g64, a1, b1, a2, b2, 0.25, round_dw=str(dtype))[0]
    errs["y@cast"], errs["dx@cast"] = err(y, y_cast, dtype), err(grads[0], dx_cast, dtype)
    bounds["y@cast"] = bounds["dx@cast"] = TOL["store_out"][dtype]
    check(f"loha_linear_full[{shape},{dtype}]", errs, bounds)


# ---- Conv2d at full size ------------------------------------------------------------------------------------------
# (B, C, H, O, k, stride): SDXL resnet conv @128, @32, downsample, 1x1 shortcut; SD1.5 1x1 proj_in, resnet conv bs 4;
# then the rest of the distinct SDXL 3x3 shapes: the up-block convs with C = 960 / 1920 / 2560
# (d = 120 / 240 / 320: other K-chunk / LDS-patch plans), the channel-changing resnet convs, both upsample convs and the
# second downsample conv
CONV = [
    (1, 320, 128, 320, 3, 1),
    (1, 1280, 32, 1280, 3, 1),
    (1, 320, 128, 320, 3, 2),
    (1, 960, 128, 320, 1, 1),
    (4, 320, 64, 320, 1, 1),
    (4, 320, 64, 320, 3, 1),
    (1, 960, 128, 320, 3, 1),
    (1, 640, 128, 320, 3, 1),
    (1, 1920, 64, 640, 3, 1),
    (1, 960, 64, 640, 3, 1),
    (1, 320, 64, 640, 3, 1),
    (1, 2560, 32, 1280, 3, 1),
    (1, 1920, 32, 1280, 3, 1),
    (1, 640, 32, 1280, 3, 1),
    (1, 640, 128, 640, 3, 1),
    (1, 1280, 64, 1280, 3, 1),
    (1, 640, 64, 640, 3, 2),
]
CIDS = [f"B{b}_{c}x{h}to{o}_k{k}s{s}" for b, c, h, o, k, s in CONV]


def _ca(k, s):
    return {"stride": s, "padding": k // 2, "dilation": 1}


@DT16
@pytest.mark.parametrize("shape", CONV, ids=CIDS)
def test_lokr_conv2d_fullsize(shape, dtype):
    from lycoris_amd import ops
    B, C, H, O, k, s = shape
    gen = torch.Generator().manual_seed(B + C + H + O + k)
    x, x64 = rnd((B, C, H, H), dtype, gen)
    Ho = (H + 2 * (k // 2) - k) // s + 1
    g, g64 = rnd((B, O, Ho, Ho), dtype, gen, 1.0 / np.sqrt(O))
    w1, w1_64 = rnd((8, 8), torch.float32, gen, 0.3)
    w2, w2_64 = rnd((O // 8, C // 8, k, k), torch.float32, gen, 0.05)
    for t in (x, w1, w2):
        t.requires_grad_(True)
    y = ops.lokr_conv2d(x, w1, w2, 1.0, (s, s), (k // 2, k // 2), (1, 1))
    dx, dw1, dw2 = torch.autograd.grad(y, [x, w1, w2], g)
    torch.cuda.synchronize()
    ca = _ca(k, s)
    y_ref = oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=1.0, kshape=(k, k), conv_args=ca)
    gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=1.0, kshape=(k, k), conv_args=ca)
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, gr["dx"], dtype), "dw1": err(dw1, gr["w1"]), "dw2": err(dw2, gr["w2"])}
    check(f"lokr_conv_full[{shape},{dtype}]", errs, _bounds(dtype, ["y", "dx"], ["dw1", "dw2"]))


@DT16
@pytest.mark.parametrize("shape", CONV, ids=CIDS)
def test_locon_conv2d_fullsize(shape, dtype):
    from lycoris_amd import ops
    B, C, H, O, k, s = shape
    r = 16 if k == 1 else 8
    gen = torch.Generator().manual_seed(B + C + H + O + k + 1)
    x, x64 = rnd((B, C, H, H), dtype, gen)
    Ho = (H + 2 * (k // 2) - k) // s + 1
    g, g64 = rnd((B, O, Ho, Ho), dtype, gen, 1.0 / np.sqrt(O))
    down, d64 = rnd((r, C, k, k), torch.float32, gen, 0.05)
    up, u64 = rnd((O, r, 1, 1), torch.float32, gen, 0.05)
    for t in (x, down, up):
        t.requires_grad_(True)
    y = ops.locon_conv2d(x, down, up, 1.0, (s, s), (k // 2, k // 2), (1, 1))
    dx, dd, du = torch.autograd.grad(y, [x, down, up], g)
    torch.cuda.synchronize()
    ca = _ca(k, s)
    y_ref = oracle.locon.forward(x64, d64, u64, 1.0, ca)
    dx_ref, dd_ref, du_ref = oracle.locon.backward(x64, g64, d64, u64, 1.0, ca)
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, dx_ref, dtype), "d_down": err(dd, dd_ref), "d_up": err(du, du_ref)}
    check(f"locon_conv_full[{shape},{dtype}]", errs, _bounds(dtype, ["y", "dx"], ["d_down", "d_up"]))


_LOHA_CONV = [0, 1, 2, 3, 4, 5, 9, 13, 16]  # + (round 4) the 320-channel convs @128, 1x1 shortcut, SD1.5 bs 4, C != O, both strides


@DT16
@pytest.mark.parametrize("shape", [CONV[i] for i in _LOHA_CONV], ids=[CIDS[i] for i in _LOHA_CONV])
def test_loha_conv2d_fullsize(shape, dtype):
    from lycoris_amd import ops
    B, C, H, O, k, s = shape
    r = 32
    gen = torch.Generator().manual_seed(B + C + H + O + k + 2)
    x, x64 = rnd((B, C, H, H), dtype, gen)
    Ho = (H + 2 * (k // 2) - k) // s + 1
    g, g64 = rnd((B, O, Ho, Ho), dtype, gen, 1.0 / np.sqrt(O))
    w1a, a1 = rnd((O, r), torch.float32, gen, 0.1)
    w1b, b1 = rnd((r, C * k * k), torch.float32, gen, 1.0)
    w2a, a2 = rnd((O, r), torch.float32, gen, 0.1)
    w2b, b2 = rnd((r, C * k * k), torch.float32, gen, 1.0)
    ts = [x, w1a, w1b, w2a, w2b]
    for t in ts:
        t.requires_grad_(True)
    y = ops.loha_conv2d(x, w1a, w1b, w2a, w2b, 0.25, (O, C, k, k), (s, s), (k // 2, k // 2), (1, 1))
    grads = torch.autograd.grad(y, ts, g)
    torch.cuda.synchronize()
    ca = _ca(k, s)
    y_ref = oracle.loha.forward(x64, a1, b1, a2, b2, 0.25, (O, C, k, k), ca)
    ref = oracle.loha.backward(x64, g64, a1, b1, a2, b2, 0.25, (O, C, k, k), ca)
    names = ["dx", "d_w1a", "d_w1b", "d_w2a", "d_w2b"]
    errs = {"y": err(y, y_ref, dtype), "dx": err(grads[0], ref[0], dtype)}
    for n, gr, rf in zip(names[1:], grads[1:], ref[1:]):
        errs[n] = err(gr, rf)
    bounds = _bounds(dtype, ["y", "dx"], names[1:], "loha_store")
    y_cast = oracle.loha.forward(x64, a1, b1, a2, b2, 0.25, (O, C, k, k), ca, round_dw=str(dtype))  # modules/loha.py:310
    dx_cast = oracle.loha.backward(x64, g64, a1, b1, a2, b2, 0.25, (O, C, k, k), ca, round_dw=str(dtype))[0]
    errs["y@cast"], errs["dx@cast"] = err(y, y_cast, dtype), err(grads[0], dx_cast, dtype)
    bounds["y@cast"] = bounds["dx@cast"] = TOL["store_out"][dtype]
    check(f"loha_conv_full[{shape},{dtype}]", errs, bounds)


# ---- low-rank LoKr (BASELINE configs[3] "(low)": dim 16, and decompose_both) at full size -------------------------
LOWRANK = [(1024, 1280, 1280), (1024, 1280, 10240), (1024, 5120, 1280), (4096, 640, 640), (4096, 640, 5120), (77, 2048, 1280)]


@DT16
@pytest.mark.parametrize("both", [False, True], ids=["w2lowrank", "decompose_both"])
@pytest.mark.parametrize("shape", LOWRANK, ids=[f"M{m}_{i}to{o}" for m, i, o in LOWRANK])
def test_lokr_lowrank_module_fullsize(shape, both, dtype):
    """LokrModule(lora_dim = 16, alpha = 8, factor = 8[, decompose_both with rank 2 on the 8 x 8 factor]) on a full-size Linear:
    w2 = w2a @ w2b (and w1 = w1a @ w1b) are products of trained factors -- reference modules/lokr.py:358-381."""
    import torch.nn as nn
    from lycoris_amd.modules import LokrModule
    M, I, O = shape
    rank = 2 if both else 16
    torch.manual_seed(M + I + O + 7)
    lin = nn.Linear(I, O, bias=False).to(dev(), dtype)
    mod = LokrModule("t", lin, 1.0, rank, rank / 2, factor=8, decompose_both=both).to(dev())
    gen = torch.Generator().manual_seed(M + I + O + 8)
    names = [n for n in ("lokr_w1", "lokr_w1_a", "lokr_w1_b", "lokr_w2", "lokr_w2_a", "lokr_w2_b") if hasattr(mod, n)]
    if both:
        assert "lokr_w1_a" in names
    assert "lokr_w2_a" in names, names
    f64 = {}
    for n in names:
        p = getattr(mod, n)
        t, t64 = rnd(tuple(p.shape), torch.float32, gen, 0.3 if "w1" in n else 0.1)
        p.data.copy_(t)
        f64[n] = t64
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
    x.requires_grad_(True)
    delta = mod.bypass_forward_diff(x) if hasattr(mod, "bypass_forward_diff") else None
    params = [getattr(mod, n) for n in names]
    grads = torch.autograd.grad(delta, [x] + params, g)
    torch.cuda.synchronize()
    kw = {"w1": f64.get("lokr_w1"), "w1a": f64.get("lokr_w1_a"), "w1b": f64.get("lokr_w1_b"),
          "w2": f64.get("lokr_w2"), "w2a": f64.get("lokr_w2_a"), "w2b": f64.get("lokr_w2_b")}
    scale = float(mod.scale)
    y_ref = oracle.lokr.forward(x64, scale=scale, **kw)
    gr = oracle.lokr.backward(x64, g64, scale=scale, **kw)
    errs = {"y": err(delta, y_ref, dtype), "dx": err(grads[0], gr["dx"], dtype)}
    for n, gv in zip(names, grads[1:]):
        errs[n] = err(gv, gr[n.replace("lokr_", "").replace("_a", "a").replace("_b", "b")])
    check(f"lokr_lowrank_full[{shape},{both},{dtype}]", errs, _bounds(dtype, ["y", "dx"], names))
