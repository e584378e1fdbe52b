"""GPU: behaviour under torch.autocast.

sd-scripts mixed precision: the frozen UNet runs under autocast, adapter parameters are fp32, and the input of e.g.
to_q/k/v is the fp32 output of a LayerNorm.  The reference's F.linear / F.conv2d autocast to the 16-bit dtype, so its
layer output is 16-bit (modules/locon.py:321-331).  The native ops must (a) return the autocast dtype -- not promote
`base + delta` to fp32 -- (b) run their 16-bit fast path, and (c) still match the oracle on the rounded input."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
from gpu_util import check, err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mods():
    from lycoris_amd.modules import IA3Module, LoConModule, LohaModule, LokrModule
    return {"locon": (LoConModule, dict(lora_dim=8, alpha=4)), "loha": (LohaModule, dict(lora_dim=4, alpha=2)),
            "lokr": (LokrModule, dict(lora_dim=100000, alpha=1, factor=8)), "ia3": (IA3Module, dict())}


def _dw64(algo, mod, layer):
    f = {n: p.detach().double().cpu().numpy() for n, p in mod.named_parameters()}
    shape = tuple(layer.weight.shape)
    if algo == "locon":
        return oracle.locon.diff_weight(f["lora_down.weight"], f["lora_up.weight"], mod.scale).reshape(shape)
    if algo == "loha":
        return oracle.loha.diff_weight(f["hada_w1_a"], f["hada_w1_b"], f["hada_w2_a"], f["hada_w2_b"], mod.scale, shape)
    if algo == "lokr":
        return oracle.lokr.diff_weight(w1=f["lokr_w1"], w2=f["lokr_w2"], scale=mod.scale, kshape=shape[2:]).reshape(shape)
    return oracle.ia3.diff_weight(layer.weight.detach().double().cpu().numpy(), f["weight"], 1.0, False)


@pytest.mark.parametrize("amp", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("algo", ["locon", "loha", "lokr", "ia3"])
@pytest.mark.parametrize("conv", [False, True], ids=["linear", "conv3x3"])
@pytest.mark.parametrize("layer_dtype", [torch.float32, "amp"], ids=["fp32_layer", "16bit_layer"])
def test_module_under_autocast(algo, conv, amp, layer_dtype):
    torch.manual_seed(7)
    ldt = amp if layer_dtype == "amp" else layer_dtype
    layer = (nn.Conv2d(64, 128, 3, padding=1) if conv else nn.Linear(64, 128)).to(DEV, ldt).requires_grad_(False)
    cls, kw = _mods()[algo]
    mod = cls("m", layer, 1.0, **kw).to(DEV)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    x = (torch.randn(2, 64, 10, 9, device=DEV) if conv else torch.randn(3, 11, 64, device=DEV)).requires_grad_(True)
    assert x.dtype == torch.float32
    mod.apply_to()
    with torch.autocast("cuda", dtype=amp):
        base_ref = mod.org_forward(x)
        out = layer(x)
        delta = mod.bypass_forward_diff(x, scale=1.0)
    mod.restore()
    assert base_ref.dtype == amp                     # what upstream's layer returns under autocast
    assert out.dtype == amp, "base + delta was promoted"
    assert delta.dtype == amp, "the adapter op did not take the 16-bit path"
    g = torch.randn_like(delta) * 0.1
    params = [p for p in mod.parameters()]
    grads = torch.autograd.grad(delta, [x] + params, g)
    assert grads[0].dtype == torch.float32            # dx goes back to the fp32 producer of x
    assert all(gr.dtype == torch.float32 for gr in grads[1:])
    # numbers: oracle on the input rounded to the autocast dtype (what the reference's autocast F.linear sees)
    x16 = x.detach().to(amp).double().cpu().numpy()
    dw = _dw64(algo, mod, layer)
    ca = {"stride": 1, "padding": 1, "dilation": 1} if conv else None
    want = oracle.general.dense_forward(x16, dw, ca)
    dx_want, _ = oracle.general.dense_backward(x16, dw, g.double().cpu().numpy(), ca)
    # IA3's delta goes through the frozen layer's autocast GEMM (W rounded to 16 bits there, output rounded once more)
    bound = (8e-3 if amp == torch.bfloat16 else 2e-3) if algo == "ia3" else 1e-3
    if algo == "loha" and amp == torch.bfloat16:
        bound = 4e-3  # dW rounded once to bf16 before the contraction, as the reference does (gpu_util.TOL["loha_store"])
    errs = {"delta": err(delta, want, amp), "dx": err(grads[0], dx_want, amp)}
    bounds = {"delta": bound, "dx": bound}
    if algo == "loha":  # paired with the north-star bound against the oracle evaluated with the reference's cast (loha.py:310)
        dwc = oracle.general.round_to(dw, str(amp))
        errs["delta@cast"] = err(delta, oracle.general.dense_forward(x16, dwc, ca), amp)
        errs["dx@cast"] = err(grads[0], oracle.general.dense_backward(x16, dwc, g.double().cpu().numpy(), ca)[0], amp)
        bounds["delta@cast"] = bounds["dx@cast"] = 1e-3
    if algo == "ia3":
        # paired with the reference's own bypass formulation and its storage roundings (modules/ia3.py:114-121): the frozen
        # layer's autocast output is a 16-bit tensor that is then scaled -- exactly the native sequence (bias-free delta)
        W16 = oracle.general.round_to(layer.weight.detach().double().cpu().numpy(), str(amp))  # autocast casts W for the GEMM
        wv = mod.weight.detach().double().cpu().numpy()
        errs["delta@bypass"] = err(delta, oracle.ia3.bypass_forward(x16, W16, wv, 1.0, False, ca, store=str(amp))[0], amp)
        errs["dx@bypass"] = err(grads[0], oracle.ia3.bypass_backward(x16, g.double().cpu().numpy(), W16, wv, 1.0, False, ca,
                                                                    store=str(amp))[0], amp)
        bounds["delta@bypass"] = bounds["dx@bypass"] = 1e-3
    check(f"autocast[{algo},{conv},{amp},{ldt}]", errs, bounds)
