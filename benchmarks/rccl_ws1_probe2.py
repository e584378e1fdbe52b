#!/usr/bin/env python3
"""Which call of the exchange blocks the HOST while the compute stream is busy?  (round 5, development)"""
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
big = torch.randn(256 << 20, device=dev)  # 1 GiB
t = torch.randn(8 << 20, device=dev)


def busy(n=20):
    for _ in range(n):
        big.mul_(1.0000001)


def us(f):
    t0 = time.perf_counter()
    f()
    return (time.perf_counter() - t0) * 1e6


def trial(tag):
    torch.cuda.synchronize()
    busy(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    busy()
    t_enq = (time.perf_counter() - t0) * 1e6
    a = us(c.mark)
    b = us(lambda: c.wait_mark(0))
    d = us(lambda: c.all_reduce(t, 1))
    e = us(c.join)
    f = us(lambda: big[:1024].add_(1.0))  # the next kernel of the compute stream
    g = us(torch.cuda.synchronize)
    print(f"{tag:34s} enqueue 20 kernels {t_enq:8.1f} us | mark {a:7.1f} wait_mark {b:7.1f} all_reduce(avg) {d:7.1f} join {e:7.1f} next kernel {f:7.1f} | drain {g:9.1f} us", flush=True)


for rep in range(2):
    trial("default (NULL) stream")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for rep in range(2):
        trial("a torch pool stream")
ev = torch.cuda.Event()
torch.cuda.synchronize()
busy()
side = torch.cuda.Stream()
ev.record()
print("torch: record", end=" ")
print(f"side.wait_event {us(lambda: side.wait_event(ev)):7.1f} us, current.wait_stream(side) {us(lambda: torch.cuda.current_stream().wait_stream(side)):7.1f} us", flush=True)
torch.cuda.synchronize()
