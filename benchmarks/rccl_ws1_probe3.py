#!/usr/bin/env python3
"""GPU-side cost of one compute -> communicator -> compute stream hop (mark, wait_mark, join), by stream kind.  (round 5, development)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lycoris_amd import _native

ext = _native.load_torch_ops()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
c = ext.RcclComm(ext.rccl_unique_id(), 0, 1, 0)
a = torch.randn(32 << 20, device=dev)  # 128 MiB: one mul_ ~ 40 us
t = torch.randn(8 << 20, device=dev)


def loop(n, hop, ar):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for _ in range(4):
            a.mul_(1.0000001)
        if hop:
            m = c.mark()
            c.wait_mark(m)
            if ar:
                c.all_reduce(t, 1)
            c.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def torch_hop(n, side):
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    t0 = time.perf_counter()
    for _ in range(n):
        for _ in range(4):
            a.mul_(1.0000001)
        side.wait_stream(cur)
        cur.wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, ctx in (("default (NULL) stream", None), ("torch pool stream", torch.cuda.Stream())):
    if ctx is not None:
        ctx.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(ctx)
    loop(20, True, True)
    base = loop(300, False, False)
    hop = loop(300, True, False)
    hop_ar = loop(300, True, True)
    th = torch_hop(300, torch.cuda.Stream())
    print(f"{name:24s}: 4 kernels {base:7.1f} us | + hop (mark, wait, join) {hop:7.1f} us | + hop with all_reduce(avg, 32 MiB) {hop_ar:7.1f} us | "
          f"+ torch side.wait_stream / cur.wait_stream {th:7.1f} us", flush=True)
