#!/usr/bin/env python3
"""A seeded sweep over the constructor arguments of the reference's LoCon / LoHa / LoKr / (IA)^3 modules on small Linear / Conv1d / Conv2d
layers: every sampled configuration is run through the REAL reference (imported read-only from /root/reference) and its forward delta,
dx and parameter gradients are stored (float64), in the layout of adapter_cases.* (make_golden.py).

    python tests/golden/make_golden_sweep.py          (build container only)

Writes tests/golden/sweep_cases.npz / sweep_cases.json.  These pin the MODULE LOGIC across the argument space (which factors exist, their
shapes, scale / alpha / rs_lora / scalar handling, Tucker, DoRA axes, unbalanced / full-matrix / decompose_both LoKr, multiplier signs) on
the host-tensor path (tests/test_golden_sweep.py); the kernels have their own parity tests against the oracle.  Configurations the
reference itself cannot run are kept in the json under "reference_fails" with its error text."""
import json
import os
import random
import sys

import numpy as np
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (toml shim + reference import + numeric_case)


def make_layer(kind, cin, cout, k=3, stride=1, padding=1, dilation=1, bias=True):
    if kind == "linear":
        return nn.Linear(cin, cout, bias=bias)
    conv = {"conv1d": nn.Conv1d, "conv2d": nn.Conv2d}[kind]
    return conv(cin, cout, k, stride, padding, dilation, bias=bias)


def sample(rng, i):
    algo = ["locon", "loha", "lokr", "lokr", "ia3"][i % 5]
    kind = rng.choice(["linear", "linear", "conv2d", "conv2d", "conv1d"])
    cin, cout = rng.choice([(12, 20), (16, 16), (24, 8), (8, 32)])
    layer = dict(kind=kind, cin=cin, cout=cout, bias=rng.random() < 0.7)
    if kind == "linear":
        xshape = rng.choice([(5, cin), (2, 3, cin)])
    else:
        k = rng.choice([1, 3, 3])
        stride = rng.choice([1, 1, 2])
        dil = rng.choice([1, 1, 2]) if k == 3 else 1
        layer.update(k=k, stride=stride, padding=(k // 2) * dil, dilation=dil)
        xshape = (2, cin, 7) if kind == "conv1d" else (2, cin, 6, 5)
    mult = rng.choice([1.0, 0.7, -0.5])
    kw = {}
    if algo == "ia3":
        kw = dict(train_on_input=rng.random() < 0.5)
    else:
        r = rng.choice([1, 2, 3, 4])
        kw = dict(lora_dim=r, alpha=rng.choice([1, 2, r, 0.5]))
        if rng.random() < 0.35:
            kw["use_tucker"] = True
        if rng.random() < 0.35:
            kw["use_scalar"] = True
        if rng.random() < 0.25:
            kw["rs_lora"] = True
        if rng.random() < 0.3:
            kw.update(weight_decompose=True, wd_on_out=rng.random() < 0.5)
        if algo == "lokr":
            kw["factor"] = rng.choice([-1, 2, 4])
            if rng.random() < 0.3:
                kw["lora_dim"] = 10000
            if rng.random() < 0.3:
                kw["decompose_both"] = True
            if rng.random() < 0.2:
                kw["full_matrix"] = True
            if rng.random() < 0.2:
                kw["unbalanced_factorization"] = True
    return algo, layer, kw, tuple(xshape), mult


def main(n=120):
    mg.make_layer = make_layer
    rng = random.Random(20250923)
    blob, metas, fails, seen = {}, {}, {}, set()
    for i in range(n):
        algo, layer_kw, mod_kw, xshape, mult = sample(rng, i)
        key = json.dumps([algo, layer_kw, mod_kw, xshape, mult], sort_keys=True)
        if key in seen:
            continue
        seen.add(key)
        name = f"sweep{i:03d}_{algo}_{layer_kw['kind']}"
        try:
            _, rec, meta = mg.numeric_case(name, algo, layer_kw, mod_kw, xshape, 1000 + i, mult)
        except Exception as e:
            fails[name] = {"algo": algo, "layer": layer_kw, "mod": mod_kw, "multiplier": mult, "error": f"{type(e).__name__}: {e}"[:240]}
            continue
        metas[name] = dict(meta, xshape=list(xshape))
        for k, v in rec.items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "sweep_cases.npz"), **blob)
    with open(os.path.join(HERE, "sweep_cases.json"), "w") as f:
        json.dump({"cases": metas, "reference_fails": fails}, f, indent=1, sort_keys=True)
    print("wrote", len(metas), "sweep cases;", len(fails), "configurations the reference cannot run")
    for k, v in fails.items():
        print("  ", k, v["mod"], "->", v["error"][:110])


if __name__ == "__main__":
    main()
