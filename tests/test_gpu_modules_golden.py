"""GPU parity of the native *modules* against the golden vectors produced by the real reference modules
(tests/golden/make_golden.py): forward delta, dx and every parameter gradient, Linear and Conv2d."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import golden_case_names
from gpu_util import check, dev, err

pytestmark = pytest.mark.gpu


def build(meta, a, dtype):
    from lycoris_amd.modules import IA3Module, LoConModule, LohaModule, LokrModule
    algos = {"locon": LoConModule, "loha": LohaModule, "lokr": LokrModule, "ia3": IA3Module}
    lk = dict(meta["layer"])
    kind = lk.pop("kind")
    bias = "bias" in a
    if kind == "linear":
        layer = nn.Linear(lk["cin"], lk["cout"], bias=bias)
    else:
        layer = nn.Conv2d(lk["cin"], lk["cout"], lk["k"], lk["stride"], lk["padding"], lk.get("dilation", 1), bias=bias)
    with torch.no_grad():
        layer.weight.copy_(torch.from_numpy(a["W"]))
        if bias:
            layer.bias.copy_(torch.from_numpy(a["bias"]))
    layer = layer.to(dev(), dtype).requires_grad_(False)
    mod = algos[meta["algo"]]("t", layer, meta["multiplier"], **meta["mod"])
    with torch.no_grad():
        for n, p in mod.named_parameters():
            p.copy_(torch.from_numpy(a["p." + n]))
    return layer, mod.to(dev())


@pytest.mark.parametrize("name", golden_case_names())
def test_module_matches_reference_golden(name, golden_cases):
    meta, a = golden_cases[name]
    dtype = torch.float32
    layer, mod = build(meta, a, dtype)
    x = torch.from_numpy(a["x"]).to(dev(), dtype).requires_grad_(True)
    g = torch.from_numpy(a["g"]).to(dev(), dtype)
    base = layer(x)
    dx_base, = torch.autograd.grad(base, x, g)
    mod.apply_to()
    out = layer(x)
    params = list(mod.named_parameters())
    grads = torch.autograd.grad(out, [x] + [p for _, p in params], g)
    mod.restore()
    assert layer.forward.__func__ is type(layer).forward  # restore() put the original forward back
    torch.cuda.synchronize()
    errs = {"delta": err(out - base, a["delta"]), "dx": err(grads[0] - dx_base, a["dx"])}
    for (n, _), gr in zip(params, grads[1:]):
        # (gtrue.*: the reference's HadaWeightTucker.backward is wrong for the a-side factors -- defect D10, make_golden.py)
        errs["g." + n] = err(gr, a.get("gtrue." + n, a["g." + n]))
    # fp32 end to end; delta/dx are differences of fp32 tensors dominated by `base`, hence the looser bound there
    bounds = {k: (2e-4 if k in ("delta", "dx") else 5e-5) for k in errs}
    check(f"module_golden[{name}]", errs, bounds)


@pytest.mark.parametrize("name", ["locon_linear", "lokr_linear_full", "loha_linear", "ia3_linear_out", "lokr_conv3_full"])
def test_module_bypass_mode_equals_rebuild_semantics(name, golden_cases):
    """bypass_mode=True must give the same numbers (rebuild semantics are canonical; upstream's LoKr
    bypass drops `scale` (D5) and its IA3 bypass scales the bias (D9) -- the native path does neither)."""
    meta, a = golden_cases[name]
    meta = dict(meta, mod=dict(meta["mod"], bypass_mode=True))
    layer, mod = build(meta, a, torch.float32)
    x = torch.from_numpy(a["x"]).to(dev(), torch.float32)
    base = layer(x)
    mod.apply_to()
    out = layer(x)
    mod.restore()
    check(f"module_bypass[{name}]", {"delta": err(out - base, a["delta"])}, {"delta": 2e-4})


def _rounded_case(a, dtype):
    """the golden case with the activations / frozen weights pre-rounded to `dtype` (what a 16-bit UNet holds); the
    adapter parameters stay fp32 (mixed-precision training)"""
    b = dict(a)
    for k in ("x", "g", "W", "bias"):
        if k in b:
            b[k] = torch.from_numpy(b[k]).to(dtype).double().numpy()
    return b


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("name", golden_case_names())
def test_module_16bit_activations_fp32_factors(name, dtype, golden_cases):
    """Modules in the mixed-precision setup (module-level tests ran in fp32 only): 16-bit frozen
    layer and activations, fp32 adapter parameters.  `base + delta` in 16 bits cannot be differenced back, so the delta
    is taken from bypass_forward_diff (the same native call forward() makes) and compared with the oracle evaluated on
    the rounded inputs."""
    from golden_util import oracle_eval
    meta, a = golden_cases[name]
    a = _rounded_case(a, dtype)
    layer, mod = build(meta, a, dtype)
    x = torch.from_numpy(a["x"]).to(dev(), dtype).requires_grad_(True)
    g = torch.from_numpy(a["g"]).to(dev(), dtype)
    dora = bool(meta["mod"].get("weight_decompose"))
    if dora:  # the DoRA delta also rescales the frozen layer's own output: hand it the 16-bit base
        delta = mod._dora_delta(x, layer(x).detach())
    else:
        delta = mod.bypass_forward_diff(x, scale=mod.multiplier)
    assert delta.dtype == dtype
    params = list(mod.named_parameters())
    grads = torch.autograd.grad(delta, [x] + [p for _, p in params], g)
    torch.cuda.synchronize()
    want_delta, want = oracle_eval(meta, a)
    # (IA)^3 goes through the frozen layer's own 16-bit GEMM (one more rounding of its output, as upstream)
    # (measured 2.6e-3 / 4.2e-3 for delta / dx in bf16: two 2^-9 roundings and the library conv's own accumulation order)
    store = (8e-3 if dtype == torch.bfloat16 else 1e-3) if meta["algo"] == "ia3" else (1e-3)
    if meta["algo"] == "loha" and dtype == torch.bfloat16:
        store = 4e-3  # dW rounded once to bf16 before the contraction, as the reference does (gpu_util.TOL["loha_store"])
    f32 = 8e-3 if meta["algo"] == "ia3" and dtype == torch.bfloat16 else (1e-3 if meta["algo"] == "ia3" else 1e-4)
    if dora:
        # the (s - 1) * (x W^T) part is computed from the frozen layer's 16-bit output (resp. goes through its 16-bit GEMM):
        # one more 2^-9 (bf16) / 2^-12 (fp16) rounding than the adapter-only deltas, as in the reference
        store = 8e-3 if dtype == torch.bfloat16 else 2e-3
        f32 = 8e-3 if dtype == torch.bfloat16 else 2e-3
    errs = {"delta": err(delta, want_delta, dtype), "dx": err(grads[0], want["dx"], dtype)}
    bounds = {"delta": store, "dx": store}
    if meta["algo"] == "loha" and not dora:
        # the loose truth-relative bf16 bound paired with the north-star bound against the cast-restating oracle (loha.py:310)
        cast_delta, cast = oracle_eval(meta, a, round_dw=str(dtype))
        errs["delta@cast"], errs["dx@cast"] = err(delta, cast_delta, dtype), err(grads[0], cast["dx"], dtype)
        bounds["delta@cast"] = bounds["dx@cast"] = 1e-3
    if meta["algo"] == "ia3":  # (the adapter delta is bias-free on both sides: rebuild semantics)
        # ... and (IA)^3 with the reference's bypass formulation + its storage roundings (modules/ia3.py:114-121)
        import oracle
        from golden_util import conv_args_of
        on_in = bool(meta["mod"].get("train_on_input", False))
        args = (a["W"], a["p.weight"], meta["multiplier"], on_in, conv_args_of(meta))
        errs["delta@bypass"] = err(delta, oracle.ia3.bypass_forward(a["x"], *args, store=str(dtype))[0], dtype)
        bdx, bdw = oracle.ia3.bypass_backward(a["x"], a["g"], *args, store=str(dtype))
        conv = meta["layer"]["kind"] != "linear"
        # Conv2d: the frozen layer's backward-data convolution (MIOpen, 16-bit) is itself 4e-3 (bf16) / 8e-4 (fp16) away from the
        # exactly-rounded product (measured: the Linear cases, whose frozen GEMM is exactly rounded, agree BIT FOR BIT), so only
        # the quantities that do not pass through it are held to the restatement
        if not conv:
            errs["dx@bypass"] = err(grads[0], bdx, dtype)
        if not (conv and on_in):
            errs["g.weight@bypass"] = err(grads[1 + [n for n, _ in params].index("weight")], bdw)
        for k in ("delta@bypass", "dx@bypass", "g.weight@bypass"):
            if k in errs:
                bounds[k] = 1e-3
    for (n, p), gr in zip(params, grads[1:]):
        assert gr.dtype == p.dtype == torch.float32
        if dora and n == "scalar":
            # d/d(scalar) under DoRA is the derivative along dW itself, which the normalisation mostly cancels: a sum of
            # O(1) terms that nets ~1e-2 of them, so 16-bit input rounding shows up 100x amplified.  Pinned in fp32 above.
            continue
        errs["g." + n] = err(gr, want["g." + n])
        bounds["g." + n] = f32
    check(f"module_16bit[{name},{dtype}]", errs, bounds)
