import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from torch.utils.checkpoint import checkpoint
from lycoris_amd import ops, _native as N
DEV = "cuda:0"
ext = N.load_torch_ops()
gen = torch.Generator(device=DEV).manual_seed(5)
f32 = dict(device=DEV, dtype=torch.float32, generator=gen)
pairs = [(nn.Parameter(torch.randn(8, 8, **f32) * 0.3), nn.Parameter(torch.randn(80, 80, **f32) * 0.05)) for _ in range(3)]
params = [p for pr in pairs for p in pr]
names = {id(p): f"p{k}{'ab'[j]}" for k, pr in enumerate(pairs) for j, p in enumerate(pr)}
x = torch.randn(128, 640, device=DEV, dtype=torch.bfloat16, generator=gen).requires_grad_(True)
g = torch.randn(128, 640, device=DEV, dtype=torch.bfloat16, generator=gen) * 0.05
def layer(h, k):
    print("   fwd layer pair", k, "state", ext.debug_recompute_state(), flush=True)
    return h + ops.lokr_linear(h, pairs[k][0], pairs[k][1], 1.0)
def block(h):
    for k in (0, 1, 0):
        h = layer(h, k)
    return h
for p in params:
    p.grad = torch.zeros_like(p)
for defer in (True, False):
    for reentrant in (False, True):
        print("defer", defer, "reentrant", reentrant, flush=True)
        ops.reset_use_counts()
        ops.deferred_weight_gradients(defer)
        ops.fused_grad_accumulation(True, callback=lambda p: print("   report", names[id(p)], flush=True))
        y = layer(checkpoint(block, x, use_reentrant=reentrant), 2)
        print("  backward", flush=True)
        y.backward(g)
        print("  backward returned", flush=True)
        ops.fused_grad_accumulation(False, None)
