#!/usr/bin/env python3
"""Golden vectors of the REAL reference on nn.Conv3d layers (FUNC_LIST / kw_dict dispatch over conv1d / 2d / 3d,
/root/reference/lycoris/functional/general.py:6, modules/base.py:89-158).

    python tests/golden/make_golden_conv3d.py          (build container only: imports /root/reference read-only)

Writes tests/golden/conv3d_cases.npz / conv3d_cases.json in the layout of adapter_cases.* (make_golden.py).  Kept in files of their own:
the Conv3d path of this repository is the ATen composite form on any device (no HIP kernel), pinned by the CPU tests only
(tests/test_conv3d.py).  Configurations the reference cannot run are listed in the json under "reference_fails" with its error."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (toml shim + reference import + numeric_case)


def conv3d_layer(kind, cin, cout, k=3, stride=1, padding=1, dilation=1, bias=True):
    assert kind == "conv3d"
    return nn.Conv3d(cin, cout, k, stride, padding, dilation, bias=bias)


def main():
    mg.make_layer = conv3d_layer
    c3 = dict(kind="conv3d", cin=8, cout=16, k=3, stride=1, padding=1)
    c3s2 = dict(kind="conv3d", cin=16, cout=8, k=3, stride=2, padding=1, bias=False)
    c1 = dict(kind="conv3d", cin=12, cout=20, k=1, stride=1, padding=0)
    c133 = dict(kind="conv3d", cin=8, cout=8, k=(1, 3, 3), stride=1, padding=(0, 1, 1))
    xc = (2, 8, 3, 4, 5)
    todo = [
        ("locon_conv3d", "locon", c3, dict(lora_dim=4, alpha=1), xc, 101),
        ("locon_conv3d_s2", "locon", c3s2, dict(lora_dim=2, alpha=2, use_scalar=True), (2, 16, 4, 5, 6), 102),
        ("locon_conv3d_k1", "locon", c1, dict(lora_dim=4, alpha=4), (2, 12, 3, 4, 5), 103),
        ("locon_conv3d_k133", "locon", c133, dict(lora_dim=2, alpha=1), (1, 8, 3, 6, 5), 104),
        ("tucker_locon_conv3d", "locon", c3, dict(lora_dim=4, alpha=1, use_tucker=True), xc, 105),
        ("dora_locon_conv3d_out", "locon", c3, dict(lora_dim=4, alpha=1, weight_decompose=True, wd_on_out=True), xc, 106),
        ("dora_locon_conv3d_in", "locon", c3, dict(lora_dim=4, alpha=1, weight_decompose=True, wd_on_out=False), xc, 107),
        ("loha_conv3d", "loha", c3, dict(lora_dim=4, alpha=1), xc, 111),
        ("loha_conv3d_s2", "loha", c3s2, dict(lora_dim=2, alpha=2, use_scalar=True), (2, 16, 4, 5, 6), 112),
        ("tucker_loha_conv3d", "loha", c3, dict(lora_dim=4, alpha=1, use_tucker=True), xc, 113),
        ("dora_loha_conv3d_out", "loha", c3, dict(lora_dim=4, alpha=1, weight_decompose=True, wd_on_out=True), xc, 114),
        ("lokr_conv3d_full", "lokr", c3, dict(lora_dim=10000, alpha=1, factor=4), xc, 121),
        ("lokr_conv3d_lowrank", "lokr", dict(kind="conv3d", cin=16, cout=32, k=3, stride=1, padding=1),
         dict(lora_dim=2, alpha=1, factor=2), (2, 16, 3, 4, 3), 122),
        ("lokr_conv3d_s2_full", "lokr", c3s2, dict(lora_dim=10000, factor=2), (2, 16, 4, 5, 6), 123),
        ("lokr_conv3d_both", "lokr", dict(kind="conv3d", cin=64, cout=96, k=1, stride=1, padding=0),
         dict(lora_dim=2, alpha=4, factor=8, decompose_both=True), (1, 64, 2, 3, 3), 124),
        ("tucker_lokr_conv3d", "lokr", dict(kind="conv3d", cin=16, cout=32, k=3, stride=1, padding=1),
         dict(lora_dim=2, alpha=1, factor=2, use_tucker=True), (2, 16, 3, 4, 3), 125),
        ("dora_lokr_conv3d_out", "lokr", c3, dict(lora_dim=10000, alpha=1, factor=4, weight_decompose=True, wd_on_out=True), xc, 126),
        ("ia3_conv3d_out", "ia3", c3, dict(), xc, 131),
        ("ia3_conv3d_in", "ia3", c3, dict(train_on_input=True), xc, 132),
    ]
    blob, metas, fails = {}, {}, {}
    for name, algo, layer_kw, mod_kw, xshape, seed in todo:
        try:
            _, rec, meta = mg.numeric_case(name, algo, layer_kw, mod_kw, xshape, seed)
        except Exception as e:  # what the reference cannot do on Conv3d is recorded, not papered over
            fails[name] = {"algo": algo, "layer": layer_kw, "mod": mod_kw, "error": f"{type(e).__name__}: {e}"[:300]}
            print("reference fails:", name, fails[name]["error"][:120])
            continue
        metas[name] = dict(meta, xshape=list(xshape))
        for k, v in rec.items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "conv3d_cases.npz"), **blob)
    with open(os.path.join(HERE, "conv3d_cases.json"), "w") as f:
        json.dump({"cases": metas, "reference_fails": fails}, f, indent=1, sort_keys=True)
    print("wrote", len(metas), "Conv3d cases;", len(fails), "configurations the reference cannot run")


if __name__ == "__main__":
    main()
