"""Round-4 parity additions (VERDICT r3 "Next round" #1):

  (b) (IA)^3 at every SDXL shape of the reference's preset (to_k / to_v on the output side, ff.net.2 on the input side),
      full size, bf16 + fp16, module level: against the float64 truth (loose: the delta passes through the frozen layer's own
      16-bit GEMM) AND against the reference's bypass formulation restated with its storage roundings (modules/ia3.py:114-121;
      oracle.ia3.bypass_*) at the north-star 1e-3;
  (d) one mixed-preset transformer block (BASELINE configs[4]: LoCon on the attention projections, LoKr on the feed-forward,
      (IA)^3 on proj_in / proj_out, fp16) trained the way bench.py trains it -- fused accumulation into the AdapterGradSync arena,
      deferred + grouped weight gradients, two micro-batches under no_sync() -- whose EVERY parameter gradient is compared with the
      oracle evaluated on the captured layer inputs / upstream gradients.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import oracle
from gpu_util import check, dev, err, rnd

pytestmark = pytest.mark.gpu

# (M, I, O, side): reference preset for (IA)^3 on SDXL (lycoris/config.py: to_k / to_v -> out side, ff.net.2 -> train_on_input)
IA3_SHAPES = [
    (1024, 1280, 1280, "out"), (4096, 640, 640, "out"), (77, 2048, 1280, "out"), (77, 2048, 640, "out"),
    (1024, 5120, 1280, "in"), (4096, 2560, 640, "in"),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", IA3_SHAPES, ids=[f"M{m}_{i}to{o}_{s}" for m, i, o, s in IA3_SHAPES])
def test_ia3_module_fullsize(shape, dtype):
    from lycoris_amd.modules import IA3Module
    M, I, O, side = shape
    on_in = side == "in"
    gen = torch.Generator().manual_seed(M + I + O)
    W, W64 = rnd((O, I), dtype, gen, 1.0 / np.sqrt(I))
    lin = nn.Linear(I, O, bias=False).to(dev(), dtype).requires_grad_(False)
    with torch.no_grad():
        lin.weight.copy_(W)
    mod = IA3Module("t", lin, 1.0, train_on_input=on_in).to(dev())
    w, w64 = rnd(tuple(mod.weight.shape), torch.float32, gen, 0.1)
    with torch.no_grad():
        mod.weight.copy_(w)
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
    x.requires_grad_(True)
    delta = mod.bypass_forward_diff(x, scale=1.0)
    dx, dw = torch.autograd.grad(delta, [x, mod.weight], g)
    torch.cuda.synchronize()
    assert delta.dtype == dtype and dw.dtype == torch.float32
    name = str(dtype)
    # truth: op(x, W * w) in float64 on the same rounded inputs (oracle.ia3 rebuild semantics)
    t_delta = oracle.ia3.forward(x64, W64, w64, 1.0, on_in)
    t_dx, t_dw = oracle.ia3.backward(x64, g64, W64, w64, 1.0, on_in)
    # the reference's bypass formulation with 16-bit intermediates (modules/ia3.py:114-121)
    b_delta, _ = oracle.ia3.bypass_forward(x64, W64, w64, 1.0, on_in, None, store=name)
    b_dx, b_dw = oracle.ia3.bypass_backward(x64, g64, W64, w64, 1.0, on_in, None, store=name)
    loose = 8e-3 if dtype == torch.bfloat16 else 1e-3
    errs = {"delta": err(delta, t_delta, dtype), "dx": err(dx, t_dx, dtype), "dw": err(dw, t_dw),
            "delta@bypass": err(delta, b_delta, dtype), "dx@bypass": err(dx, b_dx, dtype), "dw@bypass": err(dw, b_dw)}
    bounds = {"delta": loose, "dx": loose, "dw": loose, "delta@bypass": 1e-3, "dx@bypass": 1e-3, "dw@bypass": 1e-3}
    check(f"ia3_module_full[{shape},{dtype}]", errs, bounds)


# ---- (d) mixed preset block ---------------------------------------------------------------------------------------------------
class _Block(nn.Module):
    """A transformer block of the SDXL mid resolution (d = 1280, 1024 tokens, 77 context tokens of width 2048) reduced to its
    adapted projections: proj_in, attn1.{to_q,to_k,to_v,to_out}, attn2.{to_q,to_k,to_v,to_out}, ff.net.0.proj (GEGLU),
    ff.net.2, proj_out.  The attention itself is replaced by a token-mixing-free stand-in (q * sigmoid(k) + v): what matters
    here is that every adapted layer sits in ONE autograd graph, sharing inputs (q / k / v) and feeding each other."""

    def __init__(self, d=1280, ctx=2048, dtype=torch.float16):
        super().__init__()
        mk = lambda i, o, b=False: nn.Linear(i, o, bias=b)
        self.proj_in = mk(d, d, True)
        self.attn1 = nn.ModuleDict(dict(to_q=mk(d, d), to_k=mk(d, d), to_v=mk(d, d), to_out=mk(d, d, True)))
        self.attn2 = nn.ModuleDict(dict(to_q=mk(d, d), to_k=mk(ctx, d), to_v=mk(ctx, d), to_out=mk(d, d, True)))
        self.ff0 = mk(d, 8 * d, True)
        self.ff2 = mk(4 * d, d, True)
        self.proj_out = mk(d, d, True)
        self.to(dev(), dtype).requires_grad_(False)

    @staticmethod
    def _mix(a, q, k, v):
        return a["to_out"](q * torch.sigmoid(k) + v)

    def forward(self, x, ctx):
        h = self.proj_in(x)
        a = self.attn1
        h = h + self._mix(a, a["to_q"](h), a["to_k"](h), a["to_v"](h))
        a = self.attn2
        kc, vc = a["to_k"](ctx), a["to_v"](ctx)                     # [77, d]
        kq = kc.mean(0, keepdim=True).expand(h.shape[0], -1)       # stand-in for softmax(q k^T) v: broadcast the context
        vq = vc.mean(0, keepdim=True).expand(h.shape[0], -1)
        h = h + self._mix(a, a["to_q"](h), kq, vq)
        u = self.ff0(h)
        u1, u2 = u.chunk(2, dim=-1)
        h = h + self.ff2(u1 * F.gelu(u2))
        return self.proj_out(h)


def _adapted(block):
    from lycoris_amd.modules import IA3Module, LoConModule, LokrModule
    mods = {}
    for blk in ("attn1", "attn2"):
        for n, layer in getattr(block, blk).items():
            mods[f"{blk}.{n}"] = LoConModule(f"{blk}_{n}", layer, 1.0, lora_dim=16, alpha=8)
    mods["ff0"] = LokrModule("ff0", block.ff0, 1.0, lora_dim=100000, alpha=1, factor=8)
    mods["ff2"] = LokrModule("ff2", block.ff2, 1.0, lora_dim=100000, alpha=1, factor=8)
    mods["proj_in"] = IA3Module("proj_in", block.proj_in, 1.0)
    mods["proj_out"] = IA3Module("proj_out", block.proj_out, 1.0, train_on_input=True)
    return mods


def test_mixed_preset_block_every_grad_vs_oracle():
    from lycoris_amd import ops
    from lycoris_amd.grad_sync import AdapterGradSync
    dtype = torch.float16
    torch.manual_seed(11)
    block = _Block(dtype=dtype)
    with torch.no_grad():
        for p in block.parameters():  # keep the residual stream O(1) so that fp16 gradients stay well scaled
            p.mul_(0.5)
    mods = _adapted(block)
    gen = torch.Generator().manual_seed(12)
    for name, m in mods.items():
        m.to(dev())
        with torch.no_grad():
            for pn, p in m.named_parameters():
                scale = {"lora_down.weight": 0.05, "lora_up.weight": 0.05, "lokr_w1": 0.3, "lokr_w2": 0.05, "weight": 0.1}[pn]
                p.copy_((torch.randn(p.shape, generator=gen) * scale).to(p.device))
        m.apply_to()
    params = [p for m in mods.values() for p in m.parameters()]
    sync = AdapterGradSync(params, bucket_bytes=4 << 20)
    sync.attach_fused()
    ops.deferred_weight_gradients(True)
    # capture every adapted layer's input and upstream gradient (what the oracle needs), per micro-batch
    cap = {n: {"x": [], "g": []} for n in mods}
    hooks = []
    for n, m in mods.items():
        layer = m.org_module[0]
        def pre(mod_, inp, n=n):
            cap[n]["x"].append(inp[0].detach())

        def post(mod_, inp, out, n=n):  # (a forward hook's return value would REPLACE the output: return None)
            if out.requires_grad:
                out.register_hook(lambda gr, n=n: cap[n]["g"].append(gr.detach()) and None)

        hooks.append(layer.register_forward_pre_hook(pre))
        hooks.append(layer.register_forward_hook(post))
    try:
        sync.zero_grad()
        for mb in range(2):  # two micro-batches: the first under no_sync() (gradient accumulation), the second reports
            x = (torch.randn(1024, 1280, generator=gen) * 1.0).to(dev(), dtype).requires_grad_(True)
            ctx = (torch.randn(77, 2048, generator=gen)).to(dev(), dtype)
            gout = (torch.randn(1024, 1280, generator=gen) / np.sqrt(1280)).to(dev(), dtype)
            if mb == 0:
                with sync.no_sync():
                    block(x, ctx).backward(gout)
            else:
                block(x, ctx).backward(gout)
        sync.finish()
        torch.cuda.synchronize()
    finally:
        for h in hooks:
            h.remove()
        for m in mods.values():
            m.restore()
        ops.discard_deferred()  # (nothing is parked after a clean pass; a failed one must not leak into the next test)
        sync.attach_fused(False)
        sync.remove()
    errs, bounds = {}, {}
    n64 = lambda t: t.detach().double().cpu().numpy()
    for n, m in mods.items():
        xs, gs = cap[n]["x"], cap[n]["g"]
        assert len(xs) == 2 and len(gs) == 2, (n, len(xs), len(gs))
        # (one layer call per micro-batch: both lists are in micro-batch order)
        W64 = n64(m.org_module[0].weight)
        want = {}
        for x_l, g_l in zip(xs, gs):
            x64, g64 = n64(x_l), n64(g_l)
            if m.name == "locon":
                _, dd, du = oracle.locon.backward(x64, g64, n64(m.lora_down.weight), n64(m.lora_up.weight), float(m.scale))
                part = {"lora_down.weight": dd, "lora_up.weight": du}
            elif m.name == "kron":
                gr = oracle.lokr.backward(x64, g64, w1=n64(m.lokr_w1), w2=n64(m.lokr_w2), scale=float(m.scale))
                part = {"lokr_w1": gr["w1"], "lokr_w2": gr["w2"]}
            else:  # (IA)^3: the frozen layer's 16-bit output / input gradient is an intermediate (reference bypass, ia3.py:114-121)
                _, dw = oracle.ia3.bypass_backward(x64, g64, W64, n64(m.weight), 1.0, m.train_input, None, store=str(dtype))
                part = {"weight": dw}
            for k, v in part.items():
                want[k] = want.get(k, 0) + v
        for pn, p in m.named_parameters():
            key = f"{n}.{pn}"
            assert p.grad is not None and p.grad.dtype == torch.float32
            errs[key] = err(p.grad, want[pn])
            bounds[key] = 1e-3 if m.name == "ia3" else 1e-4
    check("mixed_preset_block[fp16]", errs, bounds)
