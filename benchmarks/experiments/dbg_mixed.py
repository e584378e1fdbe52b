import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import test_gpu_parity_round4 as T
from lycoris_amd import ops
from lycoris_amd.grad_sync import AdapterGradSync
from gpu_util import dev
dtype = torch.float16
torch.manual_seed(11)
block = T._Block(dtype=dtype)
mods = T._adapted(block)
gen = torch.Generator().manual_seed(12)
names = {}
for name, m in mods.items():
    m.to(dev())
    with torch.no_grad():
        for pn, p in m.named_parameters():
            p.copy_((torch.randn(p.shape, generator=gen) * 0.1).to(p.device))
            names[id(p)] = f"{name}.{pn}"
    m.apply_to()
params = [p for m in mods.values() for p in m.parameters()]
sync = AdapterGradSync(params, bucket_bytes=4 << 20)
log = []
orig = sync._on_grad_ready
def spy(p):
    log.append((names.get(id(p), "?"), sync._sync_enabled))
sync._on_grad_ready = spy
sync._on_grads_ready = lambda ps: [spy(p) for p in ps]
sync.remove()
for p in params:
    p.register_post_accumulate_grad_hook(lambda q: log.append(("HOOK " + names.get(id(q), "?"), sync._sync_enabled)))
sync.attach_fused()
ops.deferred_weight_gradients(True)
sync.zero_grad()
for mb in range(2):
    x = torch.randn(1024, 1280, generator=gen).to(dev(), dtype).requires_grad_(True)
    ctx = torch.randn(77, 2048, generator=gen).to(dev(), dtype)
    gout = (torch.randn(1024, 1280, generator=gen) / 36).to(dev(), dtype)
    log.append((f"--- mb {mb}", None))
    if mb == 0:
        with sync.no_sync():
            block(x, ctx).backward(gout)
    else:
        block(x, ctx).backward(gout)
torch.cuda.synchronize()
c = collections.Counter(n for n, e in log if e)
print([k for k, v in c.items() if v != 1], len(c))
for l in log: print(l)
