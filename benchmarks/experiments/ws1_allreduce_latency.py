"""How long does ONE in-place all-reduce of an arena slice take with RCCL at world_size 1 (nothing crosses a link)?  (round 4: the
--rccl-ws1 step pays +1.9 ms for its 5-6 collectives whatever their sizes and wherever the segments are cut.)"""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
side = torch.cuda.Stream()
for mb in (1, 2, 8, 32, 153):
    t = torch.zeros(mb * (1 << 20) // 4, device=dev)
    for mode in ("same stream", "side stream + event wait"):
        def one():
            if mode == "same stream":
                dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True).wait()
            else:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    w = dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)
                w.wait()
                torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            one()
        e1.record()
        torch.cuda.synchronize()
        print(f"{mb:4d} MiB, {mode:26s}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per all-reduce", flush=True)
dist.destroy_process_group()
