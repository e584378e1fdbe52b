"""The small host-side helpers `lycoris.functional` exports next to factorization (functional/__init__.py, general.py:6-110): FUNC_LIST,
power2factorization, tucker_weight, tucker_weight_from_conv, apply_dora_scale -- same names and argument orders here."""
import json
import os

import torch
import torch.nn.functional as F

from lycoris_amd import functional as Fn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_power2factorization_equals_the_references_on_2472_vectors():
    """golden vectors produced by the reference's function (generated in the build container, like factorization.json)"""
    with open(os.path.join(GOLDEN, "power2factorization.json")) as f:
        vec = json.load(f)
    assert len(vec) == 2472
    bad = [(d, fac, want, Fn.power2factorization(d, fac)) for d, fac, want in vec if list(Fn.power2factorization(d, fac)) != want]
    assert not bad, bad[:5]
    m, n = Fn.power2factorization(1280, 16)
    assert (m, n) == (10, 128) and m % 2 == 0 and n & (n - 1) == 0


def test_func_list_and_the_tucker_helpers():
    assert Fn.FUNC_LIST[2] is F.linear and Fn.FUNC_LIST[3:] == [F.conv1d, F.conv2d, F.conv3d] and Fn.FUNC_LIST[:2] == [None, None]
    torch.manual_seed(0)
    t, wa, wb = torch.randn(3, 4, 3, 3, dtype=torch.float64), torch.randn(3, 10, dtype=torch.float64), torch.randn(4, 6, dtype=torch.float64)
    want = torch.zeros(10, 6, 3, 3, dtype=torch.float64)
    for i in range(3):
        for j in range(4):
            want += t[i, j][None, None] * wa[i][:, None, None, None] * wb[j][None, :, None, None]
    assert torch.allclose(Fn.tucker_weight(wa, wb, t), want, atol=1e-12) and torch.allclose(Fn.rebuild_tucker(t, wa, wb), want, atol=1e-12)
    # conv-CP triple as conv weights: up [O, r, 1, 1], down [r, I, 1, 1], mid [r, r, k, k] -> the weight of the composed convolution
    up, down, mid = torch.randn(10, 3, 1, 1, dtype=torch.float64), torch.randn(4, 6, 1, 1, dtype=torch.float64), torch.randn(3, 4, 3, 3, dtype=torch.float64)
    dw = Fn.tucker_weight_from_conv(up, down, mid)
    x = torch.randn(2, 6, 5, 5, dtype=torch.float64)
    assert torch.allclose(F.conv2d(x, dw, padding=1), F.conv2d(F.conv2d(F.conv2d(x, down), mid, padding=1), up), atol=1e-12)


def test_apply_dora_scale_is_the_input_channel_decomposition():
    torch.manual_seed(1)
    W, dW = torch.randn(8, 6, 3, 3, dtype=torch.float64), torch.randn(8, 6, 3, 3, dtype=torch.float64) * 0.1
    mag = torch.rand(1, 6, 1, 1, dtype=torch.float64) + 0.5
    out = Fn.apply_dora_scale(W, dW, mag, 0.6)
    full = Fn.apply_dora_scale(W, dW, mag, 1.0)
    assert torch.allclose(full.transpose(0, 1).flatten(1).norm(dim=1), mag.flatten(), atol=1e-12)   # every input channel has its magnitude
    assert torch.allclose(out, W + (full - W) * 0.6, atol=1e-12)
    # the module path with wd_on_out=False and multiplier 1 is the same decomposition (modules/base.py _dora_merge_host, up to its eps)
    from lycoris_amd.modules import LoConModule
    layer = torch.nn.Conv2d(6, 8, 3).double()
    with torch.no_grad():
        layer.weight.copy_(W)
    m = LoConModule("t", layer, 1.0, 2, 1, weight_decompose=True, wd_on_out=False).double()
    with torch.no_grad():
        m.dora_scale.copy_(mag)
    assert torch.allclose(m._dora_merge_host(W + dW, 1.0), full, atol=1e-9)


def test_against_the_reference_when_it_is_here():
    import sys
    import types

    import pytest
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "lycoris")):
        pytest.skip("reference tree not present")
    import tomli
    shim = types.ModuleType("toml")
    shim.load = lambda f: tomli.load(open(f, "rb")) if isinstance(f, str) else tomli.load(f)
    shim.loads = tomli.loads
    sys.modules.setdefault("toml", shim)
    if ref not in sys.path:
        sys.path.insert(0, ref)
    import lycoris.functional as R
    torch.manual_seed(2)
    t, wa, wb = torch.randn(3, 4, 2, 3, 3, dtype=torch.float64), torch.randn(3, 10, dtype=torch.float64), torch.randn(4, 6, dtype=torch.float64)
    assert torch.allclose(Fn.tucker_weight(wa, wb, t), R.tucker_weight(wa, wb, t), atol=1e-12)
    up, down, mid = torch.randn(10, 3, 1, 1, dtype=torch.float64), torch.randn(4, 6, 1, 1, dtype=torch.float64), torch.randn(3, 4, 3, 3, dtype=torch.float64)
    assert torch.allclose(Fn.tucker_weight_from_conv(up, down, mid), R.tucker_weight_from_conv(up, down, mid), atol=1e-12)
    W, dW, mag = torch.randn(8, 6, dtype=torch.float64), torch.randn(8, 6, dtype=torch.float64), torch.rand(1, 6, dtype=torch.float64) + 0.5
    assert torch.allclose(Fn.apply_dora_scale(W, dW, mag, 0.3), R.apply_dora_scale(W, dW, mag, 0.3), atol=1e-12)
    assert [Fn.FUNC_LIST[i] for i in range(6)] == [R.FUNC_LIST[i] for i in range(6)]
