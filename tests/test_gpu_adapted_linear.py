"""GPU: the frozen nn.Linear and its LoKr adapter as ONE autograd node (ops.lokr_adapted_linear, round 6).

Reference: lycoris/modules/lokr.py:551-566 -- `base = org_forward(x)`, `return base + delta`: two consumers of x in the autograd graph
(F.linear and the adapter), whose input gradients the engine adds in an elementwise pass.  The owned node forms base with the library
GEMM, adds delta in the adapter kernel's epilogue and, in the backward, accumulates `g W` into the adapter's dx through the GEMM's own
epilogue.  Checked against the float64 oracle (delta and its gradients) plus `x W^T + b` / `g W` in float64, against the two-node path
on the same tensors, and through the module API (single layer and a q / k / v sibling set)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
from gpu_util import err, rnd
from lycoris_amd import ops
from lycoris_amd.modules import siblings
from lycoris_amd.modules.lokr import LokrModule

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (n layers, M rows, a, c, d, bias)
CASES = [(1, 1024, 8, 160, 160, False), (3, 1024, 8, 160, 160, True), (2, 77, 8, 160, 256, False), (1, 333, 4, 24, 40, True),
         (3, 200, 8, 80, 80, False)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", CASES, ids=[f"n{c[0]}_M{c[1]}_a{c[2]}_c{c[3]}_d{c[4]}_{'bias' if c[5] else 'nobias'}" for c in CASES])
def test_owned_node_matches_the_oracle_and_the_two_node_path(case, dtype):
    n, M, a, c, d, with_bias = case
    gen = torch.Generator().manual_seed(M + a + c + d + n)
    I, O = a * d, a * c
    x, x64 = rnd((2, M // 2, I) if M % 2 == 0 else (M, I), dtype, gen)  # [B, T, I] and plain rows
    x64 = x64.reshape(-1, I)
    Ws, bs, w1s, w2s, gs, refs = [], [], [], [], [], []
    alpha = 0.6
    for _ in range(n):
        W, W64 = rnd((O, I), dtype, gen, 1.0 / np.sqrt(I))
        b, b64 = rnd((O,), dtype, gen, 0.1) if with_bias else (None, None)
        w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
        w2, w2_64 = rnd((c, d), torch.float32, gen, 0.1)
        g, g64 = rnd((*x.shape[:-1], O), dtype, gen, 0.1)
        g64 = g64.reshape(-1, O)
        Ws.append(W); bs.append(b); w1s.append(w1.requires_grad_(True)); w2s.append(w2.requires_grad_(True)); gs.append(g)
        bw = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=alpha)
        y64 = x64 @ W64.T + (b64 if with_bias else 0.0) + oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=alpha)
        refs.append((y64, g64 @ W64 + bw["dx"], bw["w1"], bw["w2"]))
    for i in range(n):
        assert ops.lokr_linear_ownable(x, w1s[i], w2s[i], Ws[i], bs[i])
    xo = x.clone().requires_grad_(True)
    ys = ops.lokr_adapted_linear(xo, Ws, bs, w1s, w2s, [alpha] * n)
    assert len({id(y.grad_fn) for y in ys}) == 1, "one autograd node for the set"
    grads = torch.autograd.grad(ys, [xo] + w1s + w2s, gs)
    torch.cuda.synchronize()
    # the two-node path on the same tensors
    xt = x.clone().requires_grad_(True)
    bases = [torch.nn.functional.linear(xt, Ws[i], bs[i]) for i in range(n)]
    yt = ops.lokr_linear_group(xt, w1s, w2s, [alpha] * n, bases)
    gt = torch.autograd.grad(yt, [xt] + w1s + w2s, gs)
    dx_want = sum(r[1] for r in refs)
    # y: the library GEMM rounds base to the storage type, the epilogue rounds base + delta once more (the reference: base, delta and the
    # sum, three roundings): 2 x 1.7e-3 in quadrature for bf16
    tol = 3.5e-3 if dtype == torch.bfloat16 else 1e-3
    for i in range(n):
        assert err(ys[i], refs[i][0].reshape(ys[i].shape), dtype) <= tol, (i, "y")
        assert torch.equal(ys[i], yt[i]), (i, "forward bits differ from F.linear + the fused epilogue")
        assert err(grads[1 + i], refs[i][2]) <= 1e-4 and err(grads[1 + n + i], refs[i][3]) <= 1e-4, (i, "factor gradients")
    e_own, e_two = err(grads[0], dx_want.reshape(x.shape), dtype), err(gt[0], dx_want.reshape(x.shape), dtype)
    assert e_own <= (4e-3 if dtype == torch.bfloat16 else 1e-3), e_own
    assert e_own <= 1.5 * e_two + 1e-4, (e_own, e_two, "the owned node must not be less accurate than the engine's sum")


def _frozen_linear(I, O, dtype, gen, bias=True):
    layer = nn.Linear(I, O, bias=bias, device=DEV, dtype=dtype)
    with torch.no_grad():
        layer.weight.copy_((torch.randn(O, I, generator=gen) / np.sqrt(I)).to(dtype))
        if bias:
            layer.bias.copy_((torch.randn(O, generator=gen) * 0.1).to(dtype))
    layer.requires_grad_(False)
    return layer


def _lokr(layer, name, gen):
    m = LokrModule(name, layer, multiplier=1.0, lora_dim=100000, alpha=1.0, factor=8).to(DEV)
    with torch.no_grad():
        m.lokr_w1.copy_(torch.randn(m.lokr_w1.shape, generator=gen) * 0.3)
        m.lokr_w2.copy_(torch.randn(m.lokr_w2.shape, generator=gen) * 0.1)
    return m


def test_module_forward_takes_the_owned_node_and_agrees_with_the_two_node_path():
    gen = torch.Generator().manual_seed(7)
    layer = _frozen_linear(1280, 1280, torch.bfloat16, gen)
    mod = _lokr(layer, "m0", gen)
    mod.apply_to()
    x = torch.randn(2, 512, 1280, generator=gen).to(torch.bfloat16).to(DEV)
    g = (torch.randn(2, 512, 1280, generator=gen) * 0.1).to(torch.bfloat16).to(DEV)
    xa = x.clone().requires_grad_(True)
    y = layer(xa)
    assert "LokrLinearGroupFn" in type(y.grad_fn).__name__ or "LokrLinearGroupFn" in y.grad_fn.name()
    dxa, d1a, d2a = torch.autograd.grad(y, [xa, mod.lokr_w1, mod.lokr_w2], g)
    # a trainable base weight cannot be owned: the two-node path
    layer.weight.requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    yb = layer(xb)
    assert "LokrLinearGroupFn" not in yb.grad_fn.name()
    dxb, d1b, d2b = torch.autograd.grad(yb, [xb, mod.lokr_w1, mod.lokr_w2], g)
    layer.weight.requires_grad_(False)
    torch.cuda.synchronize()
    assert torch.equal(y, yb)
    assert err(dxa, dxb.double().cpu().numpy()) <= 6e-3  # bf16: different summation order of two rounded terms
    assert err(d1a, d1b.double().cpu().numpy()) <= 1e-5 and err(d2a, d2b.double().cpu().numpy()) <= 1e-5
    mod.restore()


def test_sibling_set_of_modules_runs_as_one_owned_node():
    gen = torch.Generator().manual_seed(11)
    layers = [_frozen_linear(640, 640, torch.bfloat16, gen, bias=False) for _ in range(3)]
    mods = [_lokr(l, f"s{i}", gen) for i, l in enumerate(layers)]
    for m in mods:
        m.apply_to()
    x = torch.randn(1, 256, 640, generator=gen).to(torch.bfloat16).to(DEV)
    gs = [(torch.randn(1, 256, 640, generator=gen) * 0.1).to(torch.bfloat16).to(DEV) for _ in range(3)]
    params = [p for m in mods for p in (m.lokr_w1, m.lokr_w2)]

    def run():
        xr = x.clone().requires_grad_(True)
        ys = [l(xr) for l in layers]
        return ys, torch.autograd.grad(ys, [xr] + params, gs)

    siblings.enable(False)
    try:
        ys0, gr0 = run()
    finally:
        siblings.enable(True)
    run()  # learning pass: the set forms
    ys1, gr1 = run()
    torch.cuda.synchronize()
    assert len({id(y.grad_fn) for y in ys1}) == 1 and "LokrLinearGroupFn" in ys1[0].grad_fn.name()
    for a, b in zip(ys0, ys1):
        assert torch.equal(a, b)
    assert err(gr1[0], gr0[0].double().cpu().numpy()) <= 6e-3
    for a, b in zip(gr0[1:], gr1[1:]):
        assert err(b, a.double().cpu().numpy()) <= 1e-5
    for m in mods:
        m.restore()
