"""nn.Conv1d (missing #3; reference: `FUNC_LIST` / `F.conv1d` in lycoris/functional/general.py:6, modules/base.py:89-158).

The adapter kernels are Conv2d kernels; a native module adapts an nn.Conv1d layer by being built on the layer's Conv2d TWIN
(`modules/base.py: _Conv1dTwin`: [B, C, L] <-> [B, C, 1, L], a 1 x k window, parameters shared as live views).  What must hold
and is checked here on the CPU: checkpoints have the reference's Conv1d keys and shapes, load in both directions with equal
dW, the REAL layer's forward is what gets patched / restored, the twin follows the real layer's parameters.  The numerics on the
GPU are tests/test_gpu_conv1d.py.  Conv3d layers take the ATen form (tests/test_conv3d.py)."""
import os
import sys
import types

import pytest
import torch
import torch.nn as nn

from lycoris_amd.modules import IA3Module, LoConModule, LohaModule, LokrModule, get_module, make_module

REF = "/root/reference"
CASES = [("LoConModule", dict(lora_dim=4, alpha=2)), ("LoConModule", dict(lora_dim=4, alpha=2, use_tucker=True)),
         ("LoConModule", dict(lora_dim=4, alpha=2, weight_decompose=True)), ("LohaModule", dict(lora_dim=4, alpha=2)),
         ("LohaModule", dict(lora_dim=4, alpha=2, use_tucker=True)), ("LokrModule", dict(lora_dim=100000, alpha=1, factor=8)),
         ("LokrModule", dict(lora_dim=2, alpha=1, factor=8)), ("IA3Module", dict())]
NATIVE = {"LoConModule": LoConModule, "LohaModule": LohaModule, "LokrModule": LokrModule, "IA3Module": IA3Module}


def _layer():
    torch.manual_seed(0)
    return nn.Conv1d(64, 128, 3, stride=2, padding=1)


def test_conv1d_modules_build_with_conv1d_checkpoint_shapes_and_conv3d_takes_the_aten_form():
    layer = _layer()
    m = LoConModule("t", layer, 1.0, lora_dim=4, alpha=2)
    sd = m.state_dict()
    assert tuple(sd["lora_down.weight"].shape) == (4, 64, 3) and tuple(sd["lora_up.weight"].shape) == (128, 4, 1)
    assert tuple(m.lora_down.weight.shape) == (4, 64, 1, 3)  # the native parameter: a 1 x k Conv2d window
    assert m.kw_dict["stride"] == (1, 2) and m.kw_dict["padding"] == (0, 1)
    m = LokrModule("t", layer, 1.0, lora_dim=100000, alpha=1, factor=8)
    assert tuple(m.state_dict()["lokr_w2"].shape) == (16, 8, 3)
    m3 = LoConModule("t", nn.Conv3d(8, 8, 3), 1.0, 2, 1)   # no twin, no kernel: the reference's rebuild form in ATen ops (tests/test_conv3d.py)
    assert m3._aten_only and m3._conv1d is None and tuple(m3.lora_down.weight.shape) == (2, 8, 3, 3, 3) and not m._aten_only


def test_the_real_layer_is_patched_and_the_twin_follows_its_parameters():
    layer = _layer()
    orig_forward = layer.forward
    m = LoConModule("t", layer, 1.0, lora_dim=4, alpha=2)
    twin = m.org_module[0]
    assert [n for n, _ in m.named_parameters()] == ["lora_down.weight", "lora_up.weight"]  # the frozen layer is not a submodule
    assert twin.weight.data_ptr() == layer.weight.data_ptr() and tuple(twin.weight.shape) == (128, 64, 1, 3)
    m.apply_to()
    assert layer.forward == m._forward_1d and layer.forward is not orig_forward
    # the module's view of "what is below me" is a [B, C, 1, L] function over the real 3-D forward
    x = torch.randn(2, 64, 10)
    below = m.org_forward(x.unsqueeze(2))
    assert below.shape == (2, 128, 1, 5) and torch.equal(below.squeeze(2), nn.functional.conv1d(x, layer.weight, layer.bias, 2, 1))
    m2 = LohaModule("t2", layer, 1.0, lora_dim=2, alpha=1)  # a second adapter on the same layer: the wrapper stack
    m2.apply_to()
    assert layer.forward == m2._forward_1d
    m2.restore()
    assert layer.forward == m._forward_1d
    m.restore()
    assert layer.forward.__func__ is nn.Conv1d.forward
    # parameter moves / in-place writes of the real layer are seen through the twin (no copy to keep in sync)
    layer.weight.data = layer.weight.data.double()
    assert twin.weight.dtype == torch.float64
    with torch.no_grad():
        layer.weight.zero_()
    assert float(twin.weight.detach().abs().sum()) == 0.0


def test_merge_writes_the_conv1d_weight_and_diff_weight_has_its_shape():
    layer = _layer()
    w0 = layer.weight.detach().clone()
    m = LoConModule("t", layer, 1.0, lora_dim=4, alpha=2)
    with torch.no_grad():
        m.lora_up.weight.normal_(0, 0.3)
        m.lora_down.weight.normal_(0, 0.3)
    dw = m.get_diff_weight()[0]
    assert tuple(dw.shape) == (128, 64, 3)
    want = (m.lora_up.weight.flatten(1) @ m.lora_down.weight.flatten(1)).reshape(128, 64, 3) * m.scale
    assert torch.allclose(dw, want, rtol=1e-5, atol=1e-6)
    m.merge_to(1.0)
    assert torch.allclose(layer.weight, w0 + want, rtol=1e-5, atol=1e-6)


@pytest.fixture()
def ref_modules():
    if not os.path.isdir(os.path.join(REF, "lycoris")):
        pytest.skip("reference tree not present")
    import tomli
    shim = types.ModuleType("toml")
    shim.load = lambda f: tomli.load(open(f, "rb")) if isinstance(f, str) else tomli.load(f)
    shim.loads = tomli.loads
    sys.modules.setdefault("toml", shim)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import lycoris.modules as rm
    return rm


@pytest.mark.parametrize("name,kw", CASES, ids=[f"{n}-{'-'.join(k for k in kw if k not in ('lora_dim', 'alpha'))}" for n, kw in CASES])
def test_conv1d_checkpoints_against_the_reference_modules(ref_modules, name, kw):
    layer = _layer()
    ref = getattr(ref_modules, name)("t", layer, 1.0, **kw)
    nat = NATIVE[name]("t", layer, 1.0, **kw)
    rs, ns = ref.state_dict(), nat.state_dict()
    assert {k: tuple(v.shape) for k, v in rs.items()} == {k: tuple(v.shape) for k, v in ns.items()}
    if name == "IA3Module":
        return
    # a checkpoint written by the reference loads into a native module (registry protocol) and gives the same dW, and back
    torch.manual_seed(1)
    with torch.no_grad():
        for n_, p in ref.named_parameters():
            if n_ != "dora_scale":
                p.copy_(torch.randn_like(p) * 0.3)
    sd = {f"t.{k}": v.detach().clone() for k, v in ref.state_dict().items()}
    typ, params = get_module(sd, "t")
    assert typ is NATIVE[name]
    back = make_module(typ, params, "t", layer)
    assert back is not None and back._conv1d is layer
    if kw.get("weight_decompose"):
        return  # DoRA's diff depends on W; the factor round trip is what is checked
    d_n = back.get_diff_weight()[0].detach()
    d_r = ref.get_diff_weight(1.0)
    d_r = (d_r[0] if isinstance(d_r, tuple) else d_r).detach().reshape(d_n.shape)
    if name in ("LokrModule", "LohaModule") and float(back.scale) != 1.0:
        d_r = d_r / float(back.scale)  # upstream applies `scale` twice in get_diff_weight
    assert float((d_n - d_r).norm() / d_r.norm()) <= 1e-5
    sd2 = {f"t.{k}": v.detach().clone() for k, v in back.state_dict().items()}
    typ_r, params_r = ref_modules.get_module(sd2, "t")
    ref2 = ref_modules.make_module(typ_r, params_r, "t", layer)
    assert ref2 is not None
    for k, v in ref2.state_dict().items():
        if k != "alpha":
            assert torch.equal(v, sd[f"t.{k}"]), k


def test_load_state_dict_accepts_conv1d_shaped_tensors():
    layer = _layer()
    a = LoConModule("t", layer, 1.0, lora_dim=4, alpha=2)
    with torch.no_grad():
        a.lora_up.weight.normal_(0, 0.3)
    b = LoConModule("t", layer, 1.0, lora_dim=4, alpha=2)
    b.load_state_dict(a.state_dict(), strict=False)
    assert torch.equal(b.lora_up.weight, a.lora_up.weight) and torch.equal(b.lora_down.weight, a.lora_down.weight)


def test_twin_forwards_parameter_assignments_to_the_real_layer_and_rejects_other_padding_modes():
    """`layer.bias = nn.Parameter(b)` through the twin (the merge paths on a bias-less layer) -- nn.Module.__setattr__
    would send the Parameter to register_parameter before the property setter is ever looked at; non-'zeros' padding modes are
    refused when the twin is built, as they are for nn.Conv2d."""
    from lycoris_amd.modules.base import _Conv1dTwin
    layer = nn.Conv1d(16, 24, 3, padding=1, bias=False)
    twin = _Conv1dTwin(layer)
    assert twin.bias is None
    twin.bias = nn.Parameter(torch.ones(24))
    assert layer.bias is not None and torch.equal(layer.bias, torch.ones(24)) and twin.bias is layer.bias
    twin.bias = None
    assert layer.bias is None
    w = torch.randn(24, 16, 1, 3)
    twin.weight = nn.Parameter(w)
    assert tuple(layer.weight.shape) == (24, 16, 3) and torch.equal(twin.weight, w)
    assert "bias" not in dict(twin.named_parameters()) and list(twin.parameters()) == []  # still no parameters of its own
    with pytest.raises(NotImplementedError):
        LoConModule("t", nn.Conv1d(16, 24, 3, padding=1, padding_mode="circular"), 1.0, 2, 1)
