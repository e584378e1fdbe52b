import sys, os
sys.path.insert(0, os.getcwd())
import torch, math
from torch.profiler import profile, ProfilerActivity
from lycoris_amd import ops
from lycoris_amd.grad_sync import AdapterGradSync
dev = torch.device("cuda:0")
x = torch.randn(1024, 1280, device=dev, dtype=torch.bfloat16, requires_grad=True)
g = torch.randn(1024, 1280, device=dev, dtype=torch.bfloat16)
ps = [torch.nn.Parameter(torch.randn(8, 8, device=dev) * .3), torch.nn.Parameter(torch.randn(160, 160, device=dev) * .05)]
ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
sync = AdapterGradSync(ps + ps2)
ops.fused_grad_accumulation(True)
def run():
    for pp in (ps, ps2):
        y = ops.lokr_linear(x, pp[0], pp[1], 1.0)
        torch.autograd.backward(y, g)
run(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
