#!/usr/bin/env python3
"""Development: time lyc_lokr_lr_chain_group (csrc/kron_conv.h kron_lr_chain_kernel) on the 739 nn.Linear layers of the SDXL step
at rank 16, and its two halves separately (d_w2a only / d_w2b only: the other gradient pointer NULL).

    python benchmarks/chain_bench.py [--rank 16]
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from benchmarks.sdxl_shapes import sdxl_unet_layers
from lycoris_amd import _native as N

DEV = torch.device("cuda:0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, default=16)
    args = ap.parse_args()
    r = args.rank
    N.load()
    shapes = []
    for l in sdxl_unet_layers():
        if l["kind"] == "linear":
            shapes += [(l["O"] // 8, l["I"] // 8)] * l.get("count", 1)
    keep = []
    for c, d in shapes:
        keep.append((torch.randn(c, d, device=DEV), torch.randn(c, r, device=DEV), torch.randn(r, d, device=DEV),
                     torch.zeros(c, r, device=DEV), torch.zeros(r, d, device=DEV)))

    def items(mode):
        arr = (N.LokrLrChainItem * len(shapes))()
        for k, ((c, d), (g, a, b, da, db)) in enumerate(zip(shapes, keep)):
            arr[k] = N.LokrLrChainItem(N.ptr(g), N.ptr(a), N.ptr(b), N.ptr(da) if mode != "b" else None, N.ptr(db) if mode != "a" else None,
                                       c, d, r, 1)
        return arr

    out = {"layers": len(shapes), "rank": r}
    st = torch.cuda.Stream()
    for mode in ("both", "a", "b"):
        arr = items(mode)
        with torch.cuda.stream(st):
            fn = lambda: N.call("lyc_lokr_lr_chain_group", ctypes.cast(arr, ctypes.c_void_p), len(shapes), N.stream_ptr(DEV))
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(5):
                e0.record(st)
                fn()
                e1.record(st)
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
        out[f"{mode}_ms"] = round(best, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
