"""GPU parity of the weight-space kernels (csrc/wspace.h): dW materialisation / in-place merge,
Frobenius norm for max-norm, DoRA's fused rebuild + norm and its gradient, vs the float64 oracle; then the module-level
entry points (get_diff_weight, merge_to, get_merged_weight with and without weight_decompose, apply_max_norm) against
the reference semantics restated with the oracle."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
from gpu_util import check, dev, err, rnd

pytestmark = pytest.mark.gpu


def _factors(algo, O, I, k, r, gen):
    """(device factors, float64 dW [O, I, *k] for scale 1)"""
    kk = k * k if k else 1
    ks = (k, k) if k else ()
    if algo == "locon":
        down, d64 = rnd((r, I, *ks), torch.float32, gen, 0.2)
        up, u64 = rnd((O, r, *([1] * len(ks))), torch.float32, gen, 0.2)
        return [down, up], oracle.locon.diff_weight(d64, u64, 1.0).reshape(O, I, *ks), [d64, u64]
    if algo == "loha":
        fs = [rnd((O, r), torch.float32, gen, 0.3), rnd((r, I * kk), torch.float32, gen, 0.7),
              rnd((O, r), torch.float32, gen, 0.3), rnd((r, I * kk), torch.float32, gen, 0.7)]
        n64 = [f[1] for f in fs]
        return [f[0] for f in fs], oracle.loha.diff_weight(*n64, 1.0, (O, I, *ks)), n64
    a = b = 4
    w1, n1 = rnd((a, b), torch.float32, gen, 0.4)
    w2, n2 = rnd((O // a, I // b, *ks), torch.float32, gen, 0.2)
    return [w1, w2], oracle.lokr.diff_weight(w1=n1, w2=n2, scale=1.0, kshape=ks).reshape(O, I, *ks), [n1, n2]


SHAPES = [(64, 64, 0, 8), (200, 136, 0, 16), (72, 40, 3, 4), (132, 68, 0, 40), (320, 320, 3, 8), (36, 52, 0, 3)]
IDS = [f"O{o}_I{i}_k{k}_r{r}" for o, i, k, r in SHAPES]


@pytest.mark.parametrize("algo", ["locon", "loha", "lokr"])
@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_diff_weight_merge_and_norm(algo, shape):
    from lycoris_amd import ops
    O, I, k, r = shape
    gen = torch.Generator().manual_seed(O + I + k + r)
    fs, dw64, _ = _factors(algo, O, I, k, r, gen)
    wshape = tuple(dw64.shape)
    alpha = 0.37
    errs, bounds = {}, {}
    for dt, tol in ((torch.float32, 2e-6), (torch.bfloat16, None), (torch.float16, None)):
        out = ops.diff_weight(algo, fs, wshape, alpha, dt)
        assert out.shape == wshape and out.dtype == dt
        errs[f"diff_weight[{dt}]"] = err(out, alpha * dw64, None if dt == torch.float32 else dt)
        bounds[f"diff_weight[{dt}]"] = 2e-6 if dt == torch.float32 else 1e-3  # 16-bit: rounding-boundary flips only
        W, w64 = rnd(wshape, dt, gen, 0.5)
        ops.merge_into(algo, fs, W, alpha)  # W += alpha dW in place
        errs[f"merge_into[{dt}]"] = err(W, w64 + alpha * dw64, None if dt == torch.float32 else dt)
        bounds[f"merge_into[{dt}]"] = 2e-6 if dt == torch.float32 else 1e-3
    n2 = ops.sq_norm(algo, fs, wshape, alpha)
    errs["sq_norm"] = abs(float(n2) - float((alpha * dw64).__pow__(2).sum())) / float((alpha * dw64).__pow__(2).sum())
    bounds["sq_norm"] = 1e-5
    check(f"wspace[{algo},{shape}]", errs, bounds)


@pytest.mark.parametrize("algo", ["locon", "loha", "lokr"])
@pytest.mark.parametrize("on_out", [True, False], ids=["row_norm", "col_norm"])
@pytest.mark.parametrize("shape", SHAPES[:5], ids=IDS[:5])
@pytest.mark.parametrize("wdtype", [torch.float32, torch.bfloat16], ids=["Wf32", "Wbf16"])
def test_weight_norm2_and_its_gradient(algo, on_out, shape, wdtype):
    """norm2[ch] = sum (W + alpha dW)^2 with the dW tile rebuilt on chip, and d(sum c[ch] norm2[ch]) / d factors"""
    from lycoris_amd import ops
    O, I, k, r = shape
    gen = torch.Generator().manual_seed(O + I + k + r + 7)
    fs, dw64, f64 = _factors(algo, O, I, k, r, gen)
    wshape = tuple(dw64.shape)
    W, w64 = rnd(wshape, wdtype, gen, 0.3)
    alpha = 0.6
    for f in fs:
        f.requires_grad_(True)
    mode = ops.CH_ROW if on_out else ops.CH_COL
    n2 = ops.weight_norm2(algo, W, fs, alpha, mode)
    nch = O if on_out else I
    c, c64 = rnd((nch,), torch.float32, gen, 1.0)
    grads = torch.autograd.grad(n2, fs, c)
    V = w64 + alpha * dw64
    axes = tuple(i for i in range(V.ndim) if i != (0 if on_out else 1))
    want = (V * V).sum(axis=axes)
    bs = [1] * V.ndim
    bs[0 if on_out else 1] = -1
    gV = 2.0 * V * c64.reshape(bs)  # dense gradient w.r.t. dW (times alpha inside factor_grads)
    if algo == "locon":
        ref = oracle.locon.factor_grads(gV, f64[0], f64[1], alpha)
    elif algo == "loha":
        ref = oracle.loha.factor_grads(gV, *f64, alpha)
    else:
        gr = oracle.lokr.factor_grads(gV, w1=f64[0], w2=f64[1], scale=alpha, kshape=wshape[2:])
        ref = (gr["w1"], gr["w2"])
    errs = {"norm2": err(n2, want)}
    bounds = {"norm2": 1e-5}
    for i, (g, rf) in enumerate(zip(grads, ref)):
        errs[f"g{i}"] = err(g, np.asarray(rf).reshape(g.shape))
        bounds[f"g{i}"] = 2e-5
    check(f"weight_norm2[{algo},{on_out},{shape},{wdtype}]", errs, bounds)


# ---- module level ---------------------------------------------------------------------------------------------------
def _mod(algo, layer, **kw):
    from lycoris_amd.modules import LoConModule, LohaModule, LokrModule
    cls, args = {"locon": (LoConModule, dict(lora_dim=8, alpha=4)), "loha": (LohaModule, dict(lora_dim=4, alpha=2)),
                 "lokr": (LokrModule, dict(lora_dim=100000, alpha=1, factor=4))}[algo]
    mod = cls("m", layer, 1.0, **args, **kw).to(dev())
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if n == "dora_scale":
                p.mul_(1.0 + 0.2 * torch.randn_like(p))
            elif p.dim() == 0:
                p.fill_(0.8)
            else:
                p.copy_(torch.randn_like(p) * 0.2)
    return mod


def _oracle_dw(algo, mod, shape):
    f = {n: p.detach().double().cpu().numpy() for n, p in mod.named_parameters()}
    sc = mod.scale * float(mod.scalar)  # learnable gate, or the fixed one after apply_max_norm moved it
    if algo == "locon":
        return oracle.locon.diff_weight(f["lora_down.weight"], f["lora_up.weight"], sc).reshape(shape)
    if algo == "loha":
        return oracle.loha.diff_weight(f["hada_w1_a"], f["hada_w1_b"], f["hada_w2_a"], f["hada_w2_b"], sc, shape)
    return oracle.lokr.diff_weight(w1=f["lokr_w1"], w2=f["lokr_w2"], scale=sc, kshape=tuple(shape[2:])).reshape(shape)


@pytest.mark.parametrize("algo", ["locon", "loha", "lokr"])
@pytest.mark.parametrize("conv", [False, True], ids=["linear", "conv3x3"])
@pytest.mark.parametrize("wd", [None, True, False], ids=["plain", "dora_out", "dora_in"])
def test_module_merge_paths(algo, conv, wd):
    """get_diff_weight / get_merged_weight / merge_to on the device (the tile-rebuild kernels) vs the oracle:
    merged = W + dW * mult, or with weight_decompose (W + dW) * (mult * (dora / ||W + dW|| - 1) + 1)  (locon.py:229-260)"""
    torch.manual_seed(5)
    layer = (nn.Conv2d(32, 48, 3, padding=1) if conv else nn.Linear(32, 48)).to(dev()).requires_grad_(False)
    kw = {} if wd is None else dict(weight_decompose=True, wd_on_out=wd)
    mod = _mod(algo, layer, use_scalar=True, **kw)
    shape = tuple(layer.weight.shape)
    W64 = layer.weight.detach().double().cpu().numpy()
    dw64 = _oracle_dw(algo, mod, shape)
    mult = 0.6
    diff, _ = mod.get_diff_weight(mult, shape=shape)
    merged, _ = mod.get_merged_weight(mult, shape=shape)
    if wd is None:
        want = W64 + mult * dw64
    else:
        ds = mod.dora_scale.detach().double().cpu().numpy()
        want = W64 + oracle.dora.delta_weight(W64, dw64, ds, mult, wd, eps=float(np.finfo(np.float32).eps))
    errs = {"diff": err(diff, mult * dw64), "merged": err(merged, want)}
    mod.merge_to(mult)
    errs["merge_to"] = err(layer.weight, want)
    check(f"module_merge[{algo},{conv},{wd}]", errs, {k: 5e-6 for k in errs})


@pytest.mark.parametrize("algo", ["locon", "loha", "lokr"])
def test_module_max_norm_without_dw(algo):
    """apply_max_norm (locon.py:273-284, loha.py:281-292, lokr.py:442-466): norm of dW from the factors, then the
    reference's clamp / rescale; afterwards the norm of the (oracle) dW must be the requested maximum."""
    torch.manual_seed(6)
    layer = nn.Linear(64, 96).to(dev()).requires_grad_(False)
    mod = _mod(algo, layer)
    shape = tuple(layer.weight.shape)
    n0 = float(np.linalg.norm(_oracle_dw(algo, mod, shape)))
    scaled, norm = mod.apply_max_norm(n0 * 2.0)   # above the current norm: nothing happens
    assert not scaled and abs(float(norm) - n0) / n0 < 1e-5
    scaled, norm = mod.apply_max_norm(n0 * 0.25)  # below: rescaled to the maximum
    assert scaled and abs(float(norm) - n0 * 0.25) / (n0 * 0.25) < 1e-5
    n1 = float(np.linalg.norm(_oracle_dw(algo, mod, shape)))
    assert abs(n1 - n0 * 0.25) / (n0 * 0.25) < 1e-4, (n0, n1)
