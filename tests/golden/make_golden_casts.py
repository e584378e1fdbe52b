#!/usr/bin/env python3
"""Golden vectors for the oracle's STORAGE-ROUNDING restatements, produced by the real reference run in 16 bits on the CPU.

    python tests/golden/make_golden_casts.py        (build container only: imports /root/reference)
Writes tests/golden/cast_cases.npz (+ .json).  Kept apart from make_golden.py so that adapter_cases.npz stays bit-identical.

  loha_cast_*   : `self.get_weight(self.shape).to(base_weight.dtype)` of the reference LohaModule (modules/loha.py:310) for fp32
                  factors and a 16-bit frozen layer -> pins oracle.loha.diff_weight(round_dw=...)
  ia3_bypass_*  : the reference IA3Module's bypass_forward_diff (modules/ia3.py:114-125) on a 16-bit, bias-free layer, with
                  autograd's dx / d weight -> pins oracle.ia3.bypass_forward / bypass_backward(store=...).  The input-side case
                  runs under torch.autocast (the fp32-promoted `x * weight` enters a 16-bit F.linear only there).
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

from lycoris.modules import IA3Module, LohaModule  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def loha_cast(name, layer, dt, seed):
    torch.manual_seed(seed)
    layer = layer.to(DT[dt])
    mod = LohaModule("t", layer, 1.0, lora_dim=8, alpha=4)
    with torch.no_grad():
        mod.hada_w1_a.copy_(torch.randn_like(mod.hada_w1_a) * 0.1)
        mod.hada_w1_b.copy_(torch.randn_like(mod.hada_w1_b))
        mod.hada_w2_a.copy_(torch.randn_like(mod.hada_w2_a) * 0.1)
        mod.hada_w2_b.copy_(torch.randn_like(mod.hada_w2_b))
    base_weight = layer.weight
    dw = mod.get_weight(mod.shape).to(base_weight.dtype)  # modules/loha.py:310, verbatim expression
    rec = {"dw": dw.detach().double().numpy()}
    for n, p in mod.named_parameters():
        rec["p." + n] = p.detach().double().numpy()
    meta = {"dtype": dt, "scale": float(mod.scale), "shape": list(mod.shape)}
    return name, meta, rec


def ia3_bypass(name, I, O, M, dt, on_in, seed):
    torch.manual_seed(seed)
    layer = nn.Linear(I, O, bias=False).to(DT[dt])
    mod = IA3Module("t", layer, 1.0, train_on_input=on_in)
    with torch.no_grad():
        mod.weight.copy_(torch.randn_like(mod.weight) * 0.2)
    mod.org_forward = layer.forward
    x = torch.randn(M, I).to(DT[dt]).requires_grad_(True)
    g = (torch.randn(M, O) / np.sqrt(O)).to(DT[dt])
    if on_in:
        with torch.autocast("cpu", dtype=DT[dt]):
            delta = mod.bypass_forward_diff(x, scale=1.0)
    else:
        delta = mod.bypass_forward_diff(x, scale=1.0)
    dx, dw = torch.autograd.grad(delta, [x, mod.weight], g.to(delta.dtype))
    rec = {"x": x.detach().double().numpy(), "g": g.double().numpy(), "W": layer.weight.detach().double().numpy(),
           "w": mod.weight.detach().double().numpy(), "delta": delta.detach().double().numpy(),
           "dx": dx.double().numpy(), "dw": dw.double().numpy()}
    meta = {"dtype": dt, "on_input": bool(on_in), "delta_dtype": str(delta.dtype), "dx_dtype": str(dx.dtype)}
    return name, meta, rec


def main():
    cases = []
    for dt in ("bf16", "f16"):
        cases.append(loha_cast(f"loha_cast_linear_{dt}", nn.Linear(96, 80), dt, 101))
        cases.append(loha_cast(f"loha_cast_conv3_{dt}", nn.Conv2d(24, 32, 3, padding=1), dt, 102))
        cases.append(ia3_bypass(f"ia3_bypass_out_{dt}", 160, 96, 33, dt, False, 103))
        cases.append(ia3_bypass(f"ia3_bypass_in_{dt}", 160, 96, 33, dt, True, 104))
    blob, metas = {}, {}
    for name, meta, rec in cases:
        metas[name] = meta
        for k, v in rec.items():
            blob[f"{name}/{k}"] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "cast_cases.npz"), **blob)
    with open(os.path.join(HERE, "cast_cases.json"), "w") as f:
        json.dump(metas, f, indent=1, sort_keys=True)
    print(f"wrote {len(cases)} cases, {len(blob)} arrays")


if __name__ == "__main__":
    main()
