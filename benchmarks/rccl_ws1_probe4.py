#!/usr/bin/env python3
"""Why does ONE compute -> communicator -> compute hop cost ~1 ms in the captured SDXL step and 25 us between plain kernels?
Graph replays around the hop, piece by piece.  (round 5, development)"""
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
xs = [torch.randn(1 << 20, device=dev) for _ in range(64)]
NK = int(sys.argv[1]) if len(sys.argv) > 1 else 900


def work():
    for i in range(NK):
        xs[i % 64].mul_(1.0000001)


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    work()
torch.cuda.synchronize()
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    work()
with torch.cuda.graph(g2):
    work()
tail = torch.randn(1 << 22, device=dev)


def step(variant, host):
    t0 = time.perf_counter()
    g1.replay()
    t1 = time.perf_counter()
    g2.replay()
    t2 = time.perf_counter()
    if variant in ("mark", "mark_wait", "hop"):
        m = c.mark()
        if variant != "mark":
            c.wait_mark(m)
        if variant == "hop":
            c.join()
    elif variant == "torch_event":
        e = torch.cuda.Event()
        e.record()
    tail.mul_(1.0000001)
    tail.add_(1e-9)
    t3 = time.perf_counter()
    host[0] += t1 - t0
    host[1] += t2 - t1
    host[2] += t3 - t2


for variant in ("none", "mark", "mark_wait", "hop", "torch_event", "none"):
    for _ in range(5):
        step(variant, [0, 0, 0])
    torch.cuda.synchronize()
    host = [0.0, 0.0, 0.0]
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        step(variant, host)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    print(f"{variant:12s}: {wall:7.3f} ms per step | host: first graph launch {host[0] / n * 1e6:8.1f} us, second {host[1] / n * 1e6:8.1f} us, rest {host[2] / n * 1e6:7.1f} us", flush=True)
