"""A seeded sweep over the constructor arguments of the four algorithms, pinned on the REAL reference's outputs
(tests/golden/make_golden_sweep.py: 120 sampled configurations on small Linear / Conv1d / Conv2d layers -- rank, alpha, rs_lora, scalar,
Tucker, DoRA on either axis, LoKr factor / decompose_both / full_matrix / unbalanced_factorization, (IA)^3 side, bias, stride,
dilation, negative multipliers).  Host tensors, float64: the native module classes must build the reference's parameter set (names and
shapes) and reproduce its forward delta, dx and every parameter gradient.  The kernels themselves are held to the oracle in the GPU
tests; this file holds the module logic around them to the reference across the argument space."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "sweep_cases.json")) as _f:
    _META = json.load(_f)


@pytest.fixture(scope="module")
def cases():
    blob = np.load(os.path.join(GOLDEN, "sweep_cases.npz"))
    by = {}
    for k in blob.files:
        name, arr = k.split("/", 1)
        by.setdefault(name, {})[arr] = blob[k]
    return {name: (meta, by[name]) for name, meta in _META["cases"].items()}


def _build(meta, a):
    from lycoris_amd.modules import IA3Module, LoConModule, LohaModule, LokrModule
    algos = {"locon": LoConModule, "loha": LohaModule, "lokr": LokrModule, "ia3": IA3Module}
    lk = dict(meta["layer"])
    kind = lk.pop("kind")
    if kind == "linear":
        layer = nn.Linear(lk["cin"], lk["cout"], bias=lk["bias"])
    else:
        conv = {"conv1d": nn.Conv1d, "conv2d": nn.Conv2d}[kind]
        layer = conv(lk["cin"], lk["cout"], lk["k"], lk["stride"], lk["padding"], lk["dilation"], bias=lk["bias"])
    layer = layer.double()
    with torch.no_grad():
        layer.weight.copy_(torch.from_numpy(a["W"]))
        if lk["bias"]:
            layer.bias.copy_(torch.from_numpy(a["bias"]))
    layer.requires_grad_(False)
    mod = algos[meta["algo"]]("t", layer, meta["multiplier"], **meta["mod"]).double()
    names = {n for n, _ in mod.named_parameters()}
    assert names == {k[2:] for k in a if k.startswith("p.")}, (names, sorted(k for k in a if k.startswith("p.")))
    with torch.no_grad():
        for n, p in mod.named_parameters():
            ref = a["p." + n]
            # (an nn.Conv1d adapter's native parameters are the reference's [.., .., k] tensors as [.., .., 1, k]: modules/base.py _Conv1dTwin)
            assert p.numel() == ref.size and (tuple(p.shape) == ref.shape or kind == "conv1d"), (n, tuple(p.shape), ref.shape)
            p.copy_(torch.from_numpy(ref).reshape(p.shape))
    if kind == "conv1d":  # checkpoints carry the reference's shapes
        sd = mod.state_dict()
        for n, _ in mod.named_parameters():
            if n in sd:  # (the learnable `scalar` is folded into a factor on export, like upstream's custom_state_dict)
                assert tuple(sd[n].shape) == a["p." + n].shape, n
    return layer, mod


def _err(got, want):
    got, want = got.detach().double().numpy(), np.asarray(want, dtype=np.float64)
    return float(np.linalg.norm(got - want.reshape(got.shape)) / (np.linalg.norm(want) + 1e-300))


def test_the_sweep_covers_the_argument_space_and_the_reference_ran_all_of_it():
    cs = _META["cases"]
    assert len(cs) == 120 and _META["reference_fails"] == {}
    mods = [c["mod"] for c in cs.values()]
    for key in ("use_tucker", "use_scalar", "rs_lora", "weight_decompose", "decompose_both", "full_matrix", "unbalanced_factorization",
                "train_on_input"):
        assert sum(1 for m in mods if m.get(key)) >= 3, key
    assert {c["layer"]["kind"] for c in cs.values()} == {"linear", "conv1d", "conv2d"}
    assert {c["multiplier"] for c in cs.values()} == {1.0, 0.7, -0.5}
    assert any(m.get("weight_decompose") and m.get("wd_on_out") is False for m in mods)


@pytest.mark.parametrize("name", sorted(_META["cases"]))
def test_module_matches_the_reference_on_a_sampled_configuration(name, cases):
    meta, a = cases[name]
    layer, mod = _build(meta, a)
    assert abs(float(getattr(mod, "scale", 1.0)) - meta["scale"]) < 1e-12
    x = torch.from_numpy(a["x"]).requires_grad_(True)
    g = torch.from_numpy(a["g"])
    base = layer(x)
    dx_base, = torch.autograd.grad(base, x, g)
    mod.apply_to()
    mod.train()
    out = layer(x)
    params = list(mod.named_parameters())
    grads = torch.autograd.grad(out, [x] + [p for _, p in params], g)
    mod.restore()
    errs = {"delta": _err(out - base, a["delta"]), "dx": _err(grads[0] - dx_base, a["dx"])}
    for (n, _), gr in zip(params, grads[1:]):
        errs["g." + n] = _err(gr, a.get("gtrue." + n, a["g." + n]))  # (gtrue.*: reference defect D10, make_golden.py)
    # float64 end to end; delta / dx are differences against `base` (cancellation), DoRA divides by norms
    bad = {k: v for k, v in errs.items() if v > (1e-8 if k in ("delta", "dx") else 1e-9)}
    assert not bad, (name, meta["mod"], bad)
    # bypass_mode computes the same function here (the rebuild path is canonical; upstream's bypass forms deviate from its
    # own rebuild path for LoKr -- no `scale`, D5 -- and (IA)^3 -- scaled bias, D9); with weight_decompose upstream's bypass ignores DoRA
    if not meta["mod"].get("weight_decompose"):
        mod.bypass_mode = True
        mod.apply_to()
        out_b = layer(x)
        mod.restore()
        assert _err(out_b - base, a["delta"]) < 1e-8, (name, "bypass")
        mod.bypass_mode = None
    # merging into the frozen weight reproduces the adapted forward (merge_to, modules/base.py:326-342 upstream), DoRA included
    W0 = layer.weight.detach().clone()
    mod.merge_to(meta["multiplier"])
    with torch.no_grad():
        out_m = layer(x)
        assert not torch.equal(layer.weight, W0)
        assert _err(out_m - base, a["delta"]) < 1e-8, (name, "merged")
        layer.weight.copy_(W0)


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lycoris")), reason="reference tree not present")
def test_apply_max_norm_equals_the_references_on_every_sampled_configuration(cases):
    """scale_weight_norms of sd-scripts calls Module.apply_max_norm(max_norm, device) on every adapter after each step (kohya.py
    apply_max_norm_regularization; locon.py:273-284, loha.py:281-292, lokr.py:442-466): same (scaled, norm) and the same parameters
    afterwards as the reference's module holding the same tensors -- below, inside and above the clamp window."""
    import sys
    import types

    import tomli
    shim = types.ModuleType("toml")
    shim.load = lambda f: tomli.load(open(f, "rb")) if isinstance(f, str) else tomli.load(f)
    shim.loads = tomli.loads
    sys.modules.setdefault("toml", shim)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import lycoris.modules as R
    ref_cls = {"locon": R.LoConModule, "loha": R.LohaModule, "lokr": R.LokrModule}
    checked = 0
    for name, (meta, a) in sorted(cases.items()):
        if meta["algo"] == "ia3":
            continue
        layer, mod = _build(meta, a)
        lk = dict(meta["layer"])
        kind = lk.pop("kind")
        conv = {"conv1d": nn.Conv1d, "conv2d": nn.Conv2d}.get(kind)
        rl = (nn.Linear(lk["cin"], lk["cout"], bias=lk["bias"]) if conv is None else
              conv(lk["cin"], lk["cout"], lk["k"], lk["stride"], lk["padding"], lk["dilation"], bias=lk["bias"])).double()
        with torch.no_grad():
            rl.weight.copy_(torch.from_numpy(a["W"]))
        rmod = ref_cls[meta["algo"]]("t", rl, meta["multiplier"], **meta["mod"]).double()
        with torch.no_grad():
            for n, p in rmod.named_parameters():
                p.copy_(torch.from_numpy(a["p." + n]))
        with torch.no_grad():
            norm0 = float(rmod.apply_max_norm(1e30)[1])          # (nothing is scaled at this bound: the norm itself)
        for bound in (0.5 * norm0, 1.5 * norm0, 4.0 * norm0):     # above the window, inside it, below max_norm / 2
            s_ref, n_ref = rmod.apply_max_norm(bound, None)
            s_nat, n_nat = mod.apply_max_norm(bound, None)
            assert bool(s_ref) == bool(s_nat), (name, bound)
            assert abs(float(n_ref) - float(n_nat)) <= 1e-9 * max(1.0, abs(float(n_ref))), (name, bound, float(n_ref), float(n_nat))
            for (n, p), (_, q) in zip(sorted(rmod.named_parameters()), sorted(mod.named_parameters())):
                assert _err(q, p.detach().numpy()) < 1e-10, (name, bound, n)
            if hasattr(rmod, "scalar") and not isinstance(rmod.scalar, nn.Parameter):
                assert abs(float(rmod.scalar) - float(mod.scalar)) < 1e-10, (name, bound)
        checked += 1
    assert checked >= 90
