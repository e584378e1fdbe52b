#!/usr/bin/env python3
"""Development: where the HOST time of the eager (Python-driven) adapter pass goes -- cProfile over bench.py's `nat_pass`
(forward of all 788 SDXL layers + one engine call for the backward, training configuration) -> top functions by own time.

    python benchmarks/eager_profile.py [--algo lokr]
"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]] + ["--channels-last"]
    args = bench.parse()
    bench.CHANNELS_LAST = True
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    from lycoris_amd.grad_sync import AdapterGradSync
    insts = bench.build_instances(args, torch.bfloat16, dev)
    params = [p for it in insts for p in it.params]
    sync = AdapterGradSync(params, bucket_bytes=32 << 20)
    sync.attach_fused()
    sync._sync_enabled = False

    def nat_pass():
        bench.backward_range(bench.forward_all(insts), 0, len(insts))

    for _ in range(3):
        nat_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        nat_pass()
    t_enq = (time.perf_counter() - t0) / 5 * 1e3   # host enqueue time (no sync inside)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 5 * 1e3
    # forward only / backward only, host side
    t0 = time.perf_counter()
    outs = bench.forward_all(insts)
    t_f = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    bench.backward_range(outs, 0, len(insts))
    t_b = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    print(f"layers {len(insts)}: host enqueue {t_enq:.2f} ms/pass (wall incl. GPU drain {t_all:.2f}); forward {t_f:.2f} ms host, backward {t_b:.2f} ms host")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        nat_pass()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
    print(s.getvalue()[:5000])


if __name__ == "__main__":
    main()
