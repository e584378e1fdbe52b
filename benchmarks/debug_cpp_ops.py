"""step-by-step probe of the C++ custom-op path (prints before every step; faulthandler dumps the stack on a hang)"""
import faulthandler
import os
import sys
import time

faulthandler.enable()
faulthandler.dump_traceback_later(25, repeat=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def say(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


say("import lycoris_amd")
from lycoris_amd import _native, ops
dev = torch.device("cuda:0")
x = torch.randn(64, 64, device=dev, dtype=torch.bfloat16, requires_grad=True)
w1 = (torch.randn(8, 8, device=dev) * 0.3).requires_grad_(True)
w2 = (torch.randn(16, 8, device=dev) * 0.1).requires_grad_(True)
g = torch.randn(64, 128, device=dev, dtype=torch.bfloat16)
torch.cuda.synchronize()
say("python dispatch forward")
ops.set_dispatch("python")
y0 = ops.lokr_linear(x, w1, w2, 1.0)
torch.cuda.synchronize()
say("python dispatch backward")
g0 = torch.autograd.grad(y0, [x, w1, w2], g)
torch.cuda.synchronize()
say("load torch ext")
ops.set_dispatch("cpp")
ext = _native.load_torch_ops()
say("ext loaded", ext.abi_version())
say("cpp forward, no grad")
with torch.no_grad():
    y1 = torch.ops.lycoris_amd.lokr_linear(x, w1, w2, 1.0)
torch.cuda.synchronize()
say("  equal:", torch.equal(y1, y0))
say("cpp forward, grad")
try:
    y2 = ops.lokr_linear(x, w1, w2, 1.0)
    torch.cuda.synchronize()
    say("  equal:", torch.equal(y2, y0), y2.grad_fn)
except Exception as e:
    say("  EXC", repr(e))
    raise
say("cpp backward")
try:
    g2 = torch.autograd.grad(y2, [x, w1, w2], g)
    torch.cuda.synchronize()
    say("  dx equal:", torch.equal(g2[0], g0[0]), "dw ok:", [bool(torch.allclose(a, b, rtol=1e-4, atol=1e-6)) for a, b in zip(g2[1:], g0[1:])])
except Exception as e:
    say("  EXC", repr(e))
    raise
say("locon / loha / chan")
down = (torch.randn(8, 64, device=dev) * 0.1).requires_grad_(True)
up = (torch.randn(40, 8, device=dev) * 0.1).requires_grad_(True)
y = ops.locon_linear(x, down, up, 0.5)
torch.autograd.grad(y, [x, down, up], torch.randn_like(y))
torch.cuda.synchronize()
say("  locon ok")
fs = [(torch.randn(40, 4, device=dev) * 0.3).requires_grad_(True), torch.randn(4, 64, device=dev).requires_grad_(True),
      (torch.randn(40, 4, device=dev) * 0.3).requires_grad_(True), torch.randn(4, 64, device=dev).requires_grad_(True)]
y = ops.loha_linear(x, *fs, 0.5)
torch.autograd.grad(y, [x] + fs, torch.randn_like(y))
torch.cuda.synchronize()
say("  loha ok")
w = (torch.randn(64, device=dev) * 0.3).requires_grad_(True)
y = ops.chan_affine(x, w, None, 1.0, 0.7, -1)
torch.autograd.grad(y, [x, w], torch.randn_like(y))
torch.cuda.synchronize()
say("  chan ok")
say("autocast")
xf = torch.randn(64, 64, device=dev, requires_grad=True)
with torch.autocast("cuda", dtype=torch.bfloat16):
    y = ops.lokr_linear(xf, w1, w2, 1.0)
say("  dtype", y.dtype)
gx, = torch.autograd.grad(y, [xf], torch.randn_like(y))
say("  dx dtype", gx.dtype)
say("fused accumulation + callback")
p1, p2 = torch.nn.Parameter(w1.detach().clone()), torch.nn.Parameter(w2.detach().clone())
p1.grad, p2.grad = torch.zeros_like(p1), torch.zeros_like(p2)
seen = []
ops.fused_grad_accumulation(True, callback=lambda p: seen.append(tuple(p.shape)))
y = ops.lokr_linear(x, p1, p2, 1.0)
torch.autograd.grad(y, [x], g)
torch.cuda.synchronize()
say("  seen", seen, float(p2.grad.abs().sum()))
ops.fused_grad_accumulation(False, None)
say("conv")
xc = torch.randn(2, 64, 9, 8, device=dev, dtype=torch.bfloat16, requires_grad=True)
w2c = (torch.randn(16, 8, 3, 3, device=dev) * 0.1).requires_grad_(True)
y = ops.lokr_conv2d(xc, w1, w2c, 0.7, (1, 1), (1, 1), (1, 1))
torch.autograd.grad(y, [xc, w1, w2c], torch.randn_like(y))
torch.cuda.synchronize()
say("  lokr conv ok")
dn = (torch.randn(8, 64, 3, 3, device=dev) * 0.1).requires_grad_(True)
upc = (torch.randn(48, 8, 1, 1, device=dev) * 0.1).requires_grad_(True)
y = ops.locon_conv2d(xc, dn, upc, 0.7, (1, 1), (1, 1), (1, 1))
torch.autograd.grad(y, [xc, dn, upc], torch.randn_like(y))
torch.cuda.synchronize()
say("  locon conv ok")
say("DONE")
faulthandler.cancel_dump_traceback_later()
