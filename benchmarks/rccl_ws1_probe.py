#!/usr/bin/env python3
"""What does ONE in-place collective of a 1-rank RCCL communicator cost, by stream kind and reduction op?  (round 5, development)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lycoris_amd import _native

ext = _native.load_torch_ops()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
t = torch.randn(8 << 20, device=dev)  # 32 MiB
c = ext.RcclComm(ext.rccl_unique_id(), 0, 1, 0)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize(); c.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    c.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for name, op in (("sum", 0), ("avg", 1), ("max", 2)):
    print(f"all_reduce {name:4s} 32 MiB in place: {timed(lambda: c.all_reduce(t, op)):9.1f} us per call", flush=True)
print(f"reduce_scatter sum (shard = buffer): {timed(lambda: c.reduce_scatter(t, t, 0)):9.1f} us", flush=True)
print(f"all_gather (shard = buffer):         {timed(lambda: c.all_gather(t, t)):9.1f} us", flush=True)
small = torch.randn(1024, device=dev)
print(f"all_reduce avg 4 KiB:                {timed(lambda: c.all_reduce(small, 1)):9.1f} us", flush=True)
print(f"wait_current + join only:            {timed(lambda: (c.wait_current(), c.join())):9.1f} us", flush=True)
