#!/usr/bin/env python3
"""A/B of the grouped LoKr weight-gradient launches (lyc_lokr_wgrad_group) per SDXL Linear shape: full-width tiles (kron_dw2f.h, the
default) against the round 1-3 narrow tiles (LYC_WGRAD_TILE_S), each shape in its own hipGraph, every layer its own g / x / outputs.

    python benchmarks/dw2_ab.py [--dtype bf16|f16] [--layers 24]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lycoris_amd import _native as N

DEV = torch.device("cuda:0")
SHAPES = [(1024, 1280, 1280, 372), (77, 2048, 1280, 120), (4096, 640, 640, 70), (1024, 1280, 10240, 60), (1024, 5120, 1280, 60),
          (77, 2048, 640, 20), (4096, 640, 5120, 10), (4096, 2560, 640, 10)]


def graph_us(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--factor", type=int, default=8)
    ap.add_argument("--skip-narrow", action="store_true")
    ap.add_argument("--only", type=int, default=-1, help="index into the shape table")
    ap.add_argument("--trace", action="store_true", help="-DLYC_TRACE builds: print the phase sums of workgroup 0 after the eager launches")
    ap.add_argument("--eager", type=int, default=0, help="N plain launches instead of graph timing (for rocprofv3 --pmc)")
    args = ap.parse_args()
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    code = N.dtype_code(dtype)
    a = args.factor
    tot = {"t": 0.0, "f": 0.0, "s": 0.0}
    for idx, (M, I, O, count) in enumerate(SHAPES):
        if args.only >= 0 and idx != args.only:
            continue
        c, d = O // a, I // a
        n = min(args.layers, count)
        items = (N.WgradItem * n)()
        keep = []
        for k in range(n):
            x = (torch.randn(M, I, device=DEV) * 0.5).to(dtype)
            g = (torch.randn(M, O, device=DEV) * 0.1).to(dtype)
            w1 = torch.randn(a, a, device=DEV) * 0.3
            dw2 = torch.zeros(c, d, device=DEV)
            items[k] = N.WgradItem(N.ptr(g), N.ptr(x), N.ptr(w1), None, N.ptr(dw2), None, M, a, a, c, d, 1.0)
            keep += [x, g, w1, dw2]
        tb = int(N.load().lyc_lokr_wgrad_table_bytes(n))
        table = torch.empty(tb, dtype=torch.uint8, device=DEV)
        keep.append(table)
        if args.eager:
            for _ in range(args.eager):
                N.call("lyc_lokr_wgrad_group_ws", ctypes.cast(items, ctypes.c_void_p), n, code, N.ptr(table), tb, N.stream_ptr(DEV))
            torch.cuda.synchronize()
            if args.trace:
                buf = (ctypes.c_ulonglong * 32)()
                N.load().lyc_trace_read(buf)
                ns = max(int(buf[5]), 1)
                names = ["wait DMA + barrier", "issue reads + refill, reads landed", "P wait", "mix MFMAs + hi/lo split", "main MFMAs issued"]
                print(f"  workgroup 0, {ns} steps; shader-clock ticks (100 MHz: 10 ns each) per step:")
                for nm, v in zip(names, buf[:5]):
                    print(f"    {nm:38s} {v / ns:8.2f} ticks = {v / ns * 10:7.0f} ns")
            continue
        st = torch.cuda.Stream()
        res = {}
        with torch.cuda.stream(st):
            for tag, flag in (("t", 0), ("f", 0), ("s", 0x400)):
                if tag == "s" and args.skip_narrow:
                    res[tag] = float("nan")
                    continue
                if tag == "t":
                    res[tag] = graph_us(lambda: N.call("lyc_lokr_wgrad_group_ws", ctypes.cast(items, ctypes.c_void_p), n, code, N.ptr(table), tb,
                                                       N.stream_ptr(DEV)))
                    continue
                res[tag] = graph_us(lambda: N.call("lyc_lokr_wgrad_group", ctypes.cast(items, ctypes.c_void_p), n, code | flag, N.stream_ptr(DEV)))
        nbytes = n * (M * (I + O) * 2 + c * d * 4)
        for t in res:
            tot[t] += res[t] / n * count
        print(f"M={M:5d} I={I:5d} O={O:5d} (dW2 {c}x{d}) x{n:3d}: one launch {res['t']:7.1f} us ({nbytes / res['t'] / 1e3:6.0f} GB/s)  "
              f"24/launch {res['f']:7.1f} us ({nbytes / res['f'] / 1e3:6.0f} GB/s)  narrow {res['s']:7.1f} us ({nbytes / res['s'] / 1e3:6.0f} GB/s)",
              flush=True)
    print(f"weighted over the SDXL mix: one launch {tot['t'] / 1e3:.3f} ms, 24/launch {tot['f'] / 1e3:.3f} ms, narrow {tot['s'] / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
