#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (imported read-only from /root/reference).

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py
Writes tests/golden/adapter_cases.npz   (numeric fwd/bwd pins, float64)
       tests/golden/shape_cases.json    (parameter names/shapes/scale pins for the module constructors)
       tests/golden/factorization.json  (known-answer vectors of factorization())

What is pinned (SURVEY 8c "canonical semantics"): the reference *module* rebuild path
(bypass_mode=False): delta = Module.forward(x) - org_forward(x), and autograd gradients of
sum(g * Module.forward(x)) w.r.t. every adapter parameter and (minus the frozen layer's part) x.
"""
import json
import os
import sys
import types

import numpy as np
import tomli
import torch
import torch.nn as nn

_toml = types.ModuleType("toml")
_toml.load = lambda f: tomli.load(open(f, "rb")) if isinstance(f, str) else tomli.load(f)
_toml.loads = tomli.loads
sys.modules.setdefault("toml", _toml)
sys.path.insert(0, "/root/reference")

from lycoris.functional import factorization  # noqa: E402
from lycoris.modules import IA3Module, LoConModule, LohaModule, LokrModule  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ALGOS = {"locon": LoConModule, "loha": LohaModule, "lokr": LokrModule, "ia3": IA3Module}


def make_layer(kind, cin, cout, k=3, stride=1, padding=1, dilation=1, bias=True):
    if kind == "linear":
        return nn.Linear(cin, cout, bias=bias)
    return nn.Conv2d(cin, cout, k, stride, padding, dilation, bias=bias)


def numeric_case(name, algo, layer_kw, mod_kw, xshape, seed, multiplier=0.7):
    torch.manual_seed(seed)
    layer = make_layer(**layer_kw).double()
    mod = ALGOS[algo]("t", layer, multiplier, **mod_kw).double()
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if p.dim() == 0:
                p.fill_(0.8)  # learnable `scalar`
            elif n == "dora_scale":
                p.mul_(1.0 + 0.2 * torch.randn_like(p))  # trained magnitudes: near the weight norms, not equal to them
            else:
                p.copy_(torch.randn_like(p) * 0.3)
    x = torch.randn(*xshape, dtype=torch.float64, requires_grad=True)
    base = layer(x)
    g = torch.randn_like(base)
    dx_base, = torch.autograd.grad((base * g).sum(), x)
    mod.apply_to()
    mod.train()
    out = layer(x)  # patched forward == mod.forward
    params = [(n, p) for n, p in mod.named_parameters()]
    grads = torch.autograd.grad((out * g).sum(), [x] + [p for _, p in params])
    mod.restore()
    rec = {
        "x": x.detach(), "g": g, "W": layer.weight.detach(), "delta": (out - base).detach(),
        "dx": grads[0] - dx_base,
    }
    if layer.bias is not None:
        rec["bias"] = layer.bias.detach()
    for (n, p), gr in zip(params, grads[1:]):
        rec["p." + n] = p.detach()
        rec["g." + n] = gr
    if algo == "loha" and getattr(mod, "tucker", False):
        # Reference defect D10: HadaWeightTucker.backward (functional/loha.py:49-56, 66-73) contracts grad_w with the fold of
        # the OTHER branch (`temp` is built from t2 / w2d and then used for grad_w1u, and vice versa), so its gradients of
        # hada_w1_a / hada_w2_a are not the derivative of its own forward.  The same forward through plain autograd (the
        # reference's rebuild_tucker einsum, functional/general.py:9-11) gives the true a-side gradients: stored as gtrue.*.
        from lycoris.functional.general import rebuild_tucker
        ps = dict(params)
        dw = (rebuild_tucker(ps["hada_t1"], ps["hada_w1_a"], ps["hada_w1_b"])
              * rebuild_tucker(ps["hada_t2"], ps["hada_w2_a"], ps["hada_w2_b"]) * mod.scale * ps.get("scalar", 1.0))
        if getattr(mod, "wd", False):  # (sweep cases: DoRA on top -- the reference's own decomposition of W + dW, loha.py:244-265)
            dw = mod.apply_weight_decompose(layer.weight + dw, multiplier) - layer.weight
        else:
            dw = dw * multiplier
        conv = {3: torch.nn.functional.conv1d, 4: torch.nn.functional.conv2d, 5: torch.nn.functional.conv3d}[x.dim()]
        y2 = conv(x, dw, None, layer.stride, layer.padding, layer.dilation)
        assert torch.allclose(y2, (out - base).detach(), atol=1e-10)
        ga, gb = torch.autograd.grad((y2 * g).sum(), [ps["hada_w1_a"], ps["hada_w2_a"]])
        rec["gtrue.hada_w1_a"], rec["gtrue.hada_w2_a"] = ga, gb
    meta = {
        "algo": algo, "layer": layer_kw, "mod": mod_kw, "multiplier": multiplier,
        "scale": float(getattr(mod, "scale", 1.0)),
        "scalar": float(getattr(mod, "scalar", torch.tensor(1.0))),
    }
    return name, {k: v.numpy() for k, v in rec.items()}, meta


def shape_case(algo, layer_kw, mod_kw):
    layer = make_layer(**layer_kw)
    mod = ALGOS[algo]("t", layer, 1.0, **mod_kw)
    return {
        "algo": algo, "layer": layer_kw, "mod": mod_kw,
        "params": {n: list(p.shape) for n, p in mod.named_parameters()},
        "state_dict": {n: list(v.shape) for n, v in mod.state_dict().items()},
        "scale": float(getattr(mod, "scale", 1.0)),
        "alpha": float(getattr(mod, "alpha", torch.tensor(0.0))),
        "shape": list(getattr(mod, "shape", [])),
    }


def main():
    lin = dict(kind="linear", cin=24, cout=40)
    lin_nb = dict(kind="linear", cin=32, cout=16, bias=False)
    c3 = dict(kind="conv2d", cin=8, cout=16, k=3, stride=1, padding=1)
    c3s2 = dict(kind="conv2d", cin=16, cout=8, k=3, stride=2, padding=1)
    c1 = dict(kind="conv2d", cin=12, cout=20, k=1, stride=1, padding=0)
    c3d2 = dict(kind="conv2d", cin=8, cout=8, k=3, stride=1, padding=2, dilation=2, bias=False)
    xl, xl3 = (5, 24), (2, 3, 24)
    xc = (2, 8, 6, 7)
    cases = [
        numeric_case("locon_linear", "locon", lin, dict(lora_dim=4, alpha=2), xl3, 1),
        numeric_case("locon_linear_scalar", "locon", lin_nb, dict(lora_dim=3, alpha=1, use_scalar=True), (7, 32), 2),
        numeric_case("locon_linear_rs", "locon", lin, dict(lora_dim=4, alpha=2, rs_lora=True), xl, 3, 1.0),
        numeric_case("locon_conv3", "locon", c3, dict(lora_dim=4, alpha=1), xc, 4),
        numeric_case("locon_conv3_s2", "locon", c3s2, dict(lora_dim=2, alpha=2), (2, 16, 7, 6), 5),
        numeric_case("locon_conv1", "locon", c1, dict(lora_dim=4, alpha=4), (2, 12, 5, 5), 6),
        numeric_case("locon_conv3_d2", "locon", c3d2, dict(lora_dim=2, alpha=1), (1, 8, 9, 8), 7),
        numeric_case("loha_linear", "loha", lin, dict(lora_dim=4, alpha=2), xl3, 11),
        numeric_case("loha_linear_scalar", "loha", lin_nb, dict(lora_dim=3, alpha=1, use_scalar=True), (7, 32), 12),
        numeric_case("loha_conv3", "loha", c3, dict(lora_dim=4, alpha=1), xc, 13),
        numeric_case("loha_conv3_s2", "loha", c3s2, dict(lora_dim=2, alpha=2), (2, 16, 7, 6), 14),
        numeric_case("lokr_linear_full", "lokr", lin, dict(lora_dim=10000, alpha=1, factor=4), xl3, 21),
        numeric_case("lokr_linear_full_f-1", "lokr", lin_nb, dict(lora_dim=10000, alpha=1, factor=-1), (7, 32), 22),
        numeric_case("lokr_linear_lowrank", "lokr", lin, dict(lora_dim=2, alpha=1, factor=2), xl, 23),
        numeric_case("lokr_linear_both", "lokr", dict(kind="linear", cin=64, cout=96),
                     dict(lora_dim=2, alpha=4, factor=8, decompose_both=True), (5, 64), 24),
        numeric_case("lokr_linear_scalar", "lokr", lin, dict(lora_dim=2, alpha=1, factor=4, use_scalar=True), xl, 25),
        numeric_case("lokr_linear_unbal", "lokr", lin, dict(lora_dim=10000, factor=4, unbalanced_factorization=True), xl, 26),
        numeric_case("lokr_conv3_full", "lokr", c3, dict(lora_dim=10000, alpha=1, factor=4), xc, 27),
        numeric_case("lokr_conv3_lowrank", "lokr", dict(kind="conv2d", cin=16, cout=32, k=3, stride=1, padding=1),
                     dict(lora_dim=2, alpha=1, factor=2), (2, 16, 5, 6), 28),
        numeric_case("lokr_conv3_s2_full", "lokr", c3s2, dict(lora_dim=10000, factor=2), (2, 16, 7, 6), 29),
        numeric_case("lokr_conv1_full", "lokr", c1, dict(lora_dim=10000, factor=4), (2, 12, 5, 5), 30),
        # weight_decompose (DoRA): apply_weight_decompose in the rebuild forward, both norm axes, multiplier != 1
        numeric_case("dora_locon_linear_out", "locon", lin, dict(lora_dim=4, alpha=2, weight_decompose=True, wd_on_out=True), xl3, 41),
        numeric_case("dora_locon_linear_in", "locon", lin, dict(lora_dim=4, alpha=2, weight_decompose=True, wd_on_out=False), xl, 42),
        numeric_case("dora_locon_linear_m1", "locon", lin_nb, dict(lora_dim=3, alpha=1, weight_decompose=True, wd_on_out=True, use_scalar=True), (7, 32), 43, 1.0),
        numeric_case("dora_locon_conv3_out", "locon", c3, dict(lora_dim=4, alpha=1, weight_decompose=True, wd_on_out=True), xc, 44),
        numeric_case("dora_locon_conv3_in", "locon", c3, dict(lora_dim=4, alpha=1, weight_decompose=True, wd_on_out=False), xc, 45),
        numeric_case("dora_loha_linear_out", "loha", lin, dict(lora_dim=4, alpha=2, weight_decompose=True, wd_on_out=True), xl3, 46),
        numeric_case("dora_loha_conv3_in", "loha", c3, dict(lora_dim=4, alpha=1, weight_decompose=True, wd_on_out=False), xc, 47),
        numeric_case("dora_lokr_linear_out", "lokr", lin, dict(lora_dim=10000, alpha=1, factor=4, weight_decompose=True, wd_on_out=True), xl3, 48),
        numeric_case("dora_lokr_linear_in_lowrank", "lokr", lin, dict(lora_dim=2, alpha=1, factor=2, weight_decompose=True, wd_on_out=False), xl, 49),
        numeric_case("dora_lokr_conv3_out", "lokr", c3, dict(lora_dim=10000, alpha=1, factor=4, weight_decompose=True, wd_on_out=True), xc, 50),
        # Tucker / conv-CP forms (use_tucker on k > 1 convolutions): lora_mid, HadaWeightTucker, lokr_t2
        numeric_case("tucker_locon_conv3", "locon", c3, dict(lora_dim=4, alpha=1, use_tucker=True), xc, 61),
        numeric_case("tucker_locon_conv3_s2", "locon", c3s2, dict(lora_dim=2, alpha=2, use_tucker=True, use_scalar=True), (2, 16, 7, 6), 62),
        numeric_case("tucker_loha_conv3", "loha", c3, dict(lora_dim=4, alpha=1, use_tucker=True), xc, 63),
        numeric_case("tucker_lokr_conv3", "lokr", dict(kind="conv2d", cin=16, cout=32, k=3, stride=1, padding=1),
                     dict(lora_dim=2, alpha=1, factor=2, use_tucker=True), (2, 16, 5, 6), 64),
        numeric_case("tucker_lokr_conv3_d2", "lokr", dict(kind="conv2d", cin=16, cout=16, k=3, stride=1, padding=2, dilation=2),
                     dict(lora_dim=2, alpha=2, factor=4, use_tucker=True), (1, 16, 9, 8), 65),
        numeric_case("ia3_linear_out", "ia3", lin, dict(), xl3, 31),
        numeric_case("ia3_linear_in", "ia3", lin, dict(train_on_input=True), xl3, 32),
        numeric_case("ia3_linear_out_nobias", "ia3", lin_nb, dict(), (7, 32), 33),
        numeric_case("ia3_conv3_out", "ia3", c3, dict(), xc, 34),
        numeric_case("ia3_conv3_in", "ia3", c3, dict(train_on_input=True), xc, 35),
    ]
    blob, metas = {}, {}
    for name, rec, meta in cases:
        metas[name] = meta
        for k, v in rec.items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "adapter_cases.npz"), **blob)
    with open(os.path.join(HERE, "adapter_cases.json"), "w") as f:
        json.dump(metas, f, indent=1, sort_keys=True)

    shapes = []
    lin_dims = [(320, 320), (1280, 10240), (5120, 1280), (2048, 640), (768, 320), (250, 360), (127, 64), (64, 96)]
    for cin, cout in lin_dims:
        lk = dict(kind="linear", cin=cin, cout=cout)
        shapes.append(shape_case("locon", lk, dict(lora_dim=16, alpha=8)))
        shapes.append(shape_case("loha", lk, dict(lora_dim=8, alpha=4)))
        shapes.append(shape_case("ia3", lk, dict()))
        shapes.append(shape_case("ia3", lk, dict(train_on_input=True)))
        for factor in (-1, 4, 8, 16):
            for dim, extra in ((4, {}), (16, {}), (10000, {}), (4, dict(decompose_both=True)),
                               (2, dict(decompose_both=True)),
                               (4, dict(full_matrix=True)), (8, dict(unbalanced_factorization=True)),
                               (8, dict(rs_lora=True, alpha=2))):
                kw = dict(lora_dim=dim, alpha=extra.pop("alpha", 1), factor=factor, **extra)
                shapes.append(shape_case("lokr", lk, kw))
    for cin, cout, k in [(320, 320, 3), (640, 1280, 3), (1920, 640, 1), (8, 16, 3), (4, 320, 3)]:
        ck = dict(kind="conv2d", cin=cin, cout=cout, k=k, stride=1, padding=k // 2)
        shapes.append(shape_case("locon", ck, dict(lora_dim=8, alpha=4)))
        shapes.append(shape_case("loha", ck, dict(lora_dim=8, alpha=4)))
        shapes.append(shape_case("ia3", ck, dict()))
        for factor in (-1, 8):
            for dim, extra in ((4, {}), (10000, {}), (2, dict(decompose_both=True)), (4, dict(full_matrix=True))):
                shapes.append(shape_case("lokr", ck, dict(lora_dim=dim, alpha=1, factor=factor, **extra)))
    with open(os.path.join(HERE, "shape_cases.json"), "w") as f:
        json.dump(shapes, f, sort_keys=True)

    fac = [[d, f, list(factorization(d, f))]
           for d in list(range(1, 400)) + [512, 640, 768, 1024, 1280, 2048, 2560, 5120, 10240]
           for f in (-1, 2, 4, 8, 12, 16)]
    with open(os.path.join(HERE, "factorization.json"), "w") as f:
        json.dump(fac, f)
    print("wrote", len(cases), "numeric cases,", len(shapes), "shape cases,", len(fac), "factorization vectors")


if __name__ == "__main__":
    main()
