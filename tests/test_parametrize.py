"""torch parametrize integration (reference modules/base.py:199-234, 392-395; docs/API.md lists `parametrize` / `parametrize_forward` in
the module API; reference test/module.py:149-190 smoke-tests it): `Module.parametrize(layer, "weight", ...)` registers the adapter as a
parametrization, reading `layer.weight` then yields the merged weight, differentiable w.r.t. the adapter's parameters."""
import os
import sys
import types

import pytest
import torch
import torch.nn as nn
import torch.nn.utils.parametrize as P

from lycoris_amd.modules import IA3Module, LoConModule, LohaModule, LokrModule

REF = "/root/reference"
NATIVE = {"LoConModule": LoConModule, "LohaModule": LohaModule, "LokrModule": LokrModule, "IA3Module": IA3Module}
VARIANTS = [("LoConModule", {}), ("LoConModule", dict(weight_decompose=True)), ("LoConModule", dict(use_tucker=True, use_scalar=True)),
            ("LohaModule", {}), ("LohaModule", dict(use_tucker=True)), ("LohaModule", dict(weight_decompose=True, use_scalar=True)),
            ("LokrModule", dict(factor=4)), ("LokrModule", dict(factor=4, use_tucker=True)), ("LokrModule", dict(factor=4, weight_decompose=True)),
            ("IA3Module", {})]


def _layers():
    torch.manual_seed(0)
    return {"linear": (nn.Linear(16, 16), torch.randn(3, 16)), "conv2d": (nn.Conv2d(16, 16, 3, padding=1), torch.randn(2, 16, 5, 4)),
            "conv1d": (nn.Conv1d(16, 16, 3, padding=1), torch.randn(2, 16, 7)), "conv3d": (nn.Conv3d(16, 16, 3, padding=1), torch.randn(1, 16, 3, 4, 3))}


def _randomise(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in sorted(mod.named_parameters()):
            if n.endswith("original"):
                continue
            if n == "dora_scale":
                p.mul_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 0:
                p.fill_(0.8)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)


@pytest.mark.parametrize("kind", ["linear", "conv2d", "conv1d", "conv3d"])
@pytest.mark.parametrize("cls_name,kw", VARIANTS, ids=[f"{c}-{'-'.join(k) or 'plain'}" for c, k in VARIANTS])
def test_parametrized_weight_is_the_merged_weight_and_trains(cls_name, kw, kind):
    layer, x = _layers()[kind]
    W0 = layer.weight.detach().clone()
    cls = NATIVE[cls_name]
    mod = cls.parametrize(layer, "weight", 0.7, 4, 1, **kw)
    assert P.is_parametrized(layer, "weight") and layer.parametrizations.weight[0] is mod and mod.bypass_mode is False
    _randomise(mod, 1)
    # reading the attribute gives the merged weight of the module's own weight-space API
    shape = tuple(mod.shape)
    if cls_name == "IA3Module":
        want = mod.get_merged_weight(0.7, shape)[0]
    elif kw.get("weight_decompose"):
        want = mod._dora_merge_host(W0.reshape(shape) + mod.get_diff_weight(1.0, shape)[0], 0.7)
    else:
        want = W0.reshape(shape) + mod.get_diff_weight(0.7, shape)[0]
    got = layer.weight
    assert got.shape == W0.shape and got.requires_grad
    assert torch.allclose(got, want.reshape(W0.shape).detach(), atol=1e-6) and not torch.allclose(got.detach(), W0, atol=1e-4)
    # forward through the adapted layer, gradients reach every adapter parameter, the original weight is still the stored tensor
    y = layer(x)
    y.square().sum().backward()
    for n, p in mod.named_parameters():
        assert p.grad is not None and float(p.grad.abs().sum()) > 0, n
    assert torch.equal(layer.parametrizations.weight.original.detach(), W0)
    # state_dict round trip (reference test/module.py:183-184)
    mod.load_state_dict(mod.state_dict())
    assert torch.allclose(layer.weight.detach(), got.detach(), atol=1e-6)
    P.remove_parametrizations(layer, "weight", leave_parametrized=False)
    assert torch.equal(layer.weight.detach(), W0)


def test_non_square_targets_keep_the_weights_orientation():
    """the reference builds its proxy with (shape[0], shape[1]) in the (in, out) slots (base.py:209-229) and so only handles square
    weights; here the proxy has the weight's own orientation"""
    layer = nn.Linear(24, 40)
    mod = LokrModule.parametrize(layer, "weight", 1.0, 10000, 1, factor=4)
    with torch.no_grad():
        mod.lokr_w2.normal_(std=0.1)
    assert tuple(mod.shape) == (40, 24) and tuple(layer.weight.shape) == (40, 24)
    want = layer.parametrizations.weight.original + mod.get_diff_weight(1.0, (40, 24))[0]
    assert torch.allclose(layer.weight, want, atol=1e-6)
    conv = nn.Conv2d(8, 12, 3)
    m2 = LoConModule.parametrize(conv, "weight", 1.0, 2, 1)
    assert tuple(m2.shape) == (12, 8, 3, 3) and conv(torch.randn(1, 8, 5, 5)).shape == (1, 12, 3, 3)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lycoris")), reason="reference tree not present")
@pytest.mark.parametrize("kind", ["linear", "conv2d"])
@pytest.mark.parametrize("cls_name,kw", VARIANTS, ids=[f"{c}-{'-'.join(k) or 'plain'}" for c, k in VARIANTS])
def test_parametrized_weight_equals_the_references(cls_name, kw, kind):
    import tomli
    shim = types.ModuleType("toml")
    shim.load = lambda f: tomli.load(open(f, "rb")) if isinstance(f, str) else tomli.load(f)
    shim.loads = tomli.loads
    sys.modules.setdefault("toml", shim)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import lycoris.modules as ref_modules

    # alpha = rank (scale 1) and a unit `scalar` for LoHa / LoKr: the reference's get_diff_weight of these two applies `scale` twice and
    # leaves the `scalar` gate out (reference defect D7, SURVEY 8c: loha.py:228-230 with :206, lokr.py:383-385 with :370), and its
    # parametrization goes through that function.  The trained forward uses scale and scalar once; so does this repository everywhere,
    # so the two agree exactly where the defect is invisible -- which is what is compared here.
    unit_scalar = cls_name in ("LohaModule", "LokrModule")

    def run(cls):
        layer, x = _layers()[kind]
        layer, x = layer.double(), x.double()
        mod = cls.parametrize(layer, "weight", 0.7, 4, 4, **kw).double()
        _randomise(mod, 1)
        if unit_scalar and isinstance(getattr(mod, "scalar", None), nn.Parameter):
            with torch.no_grad():
                mod.scalar.fill_(1.0)
        w = layer.weight
        y = layer(x)
        params = [(n, p) for n, p in sorted(mod.named_parameters())]
        grads = torch.autograd.grad(y.square().sum(), [p for _, p in params], allow_unused=True)
        return w.detach(), y.detach(), {n: g for (n, _), g in zip(params, grads)}

    w_ref, y_ref, g_ref = run(getattr(ref_modules, cls_name))
    w_nat, y_nat, g_nat = run(NATIVE[cls_name])
    rel = lambda a, b: float((a - b.reshape(a.shape)).norm() / (b.norm() + 1e-300))
    assert rel(w_nat, w_ref) < 1e-12 and rel(y_nat, y_ref) < 1e-12, (rel(w_nat, w_ref), rel(y_nat, y_ref))
    assert set(g_nat) == set(g_ref)
    for n in g_ref:
        if cls_name == "LohaModule" and kw.get("use_tucker") and n in ("hada_w1_a", "hada_w2_a"):
            continue  # reference defect D10 (HadaWeightTucker.backward), tests/golden/make_golden.py
        if g_ref[n] is None:
            assert unit_scalar and n == "scalar"  # (the gate the reference leaves out of its weight-space functions)
            continue
        assert rel(g_nat[n], g_ref[n]) < 1e-10, (n, rel(g_nat[n], g_ref[n]))
