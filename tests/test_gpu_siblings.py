"""GPU: sibling projections behind the module API (round 5; VERDICT r4 next #4a).

  * torch.ops.lycoris_amd.lokr_linear_group (one forward launch, one autograd node with n outputs, one backward dx launch, parked
    weight gradients) against the float64 oracle and, bit for bit, against n lokr_linear calls;
  * stock attention-shaped host code (to_q(h), to_k(ctx), to_v(ctx)) adapted by LokrModule: the modules find their sibling sets on
    the first forward pass (modules/siblings.py) and from the second pass on run them as grouped launches -- same bits in the
    outputs, same gradients, fewer launches.
Reference call sites: one LokrModule.forward per projection, lycoris/modules/lokr.py:543-566."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
from gpu_util import TOL, check, err, rnd
from lycoris_amd import ops
from lycoris_amd.modules import LokrModule, siblings

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

# (M, a, c, d, n): q / k / v of a 1280-wide SDXL block, k / v of its cross-attention, a 640-wide block, ragged small case
CASES = [(1024, 8, 160, 160, 3), (77, 8, 160, 256, 2), (4096, 8, 80, 80, 3), (130, 4, 24, 40, 4)]


@pytest.mark.parametrize("with_base", [False, True], ids=["delta", "base_plus_delta"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", CASES, ids=[f"M{c[0]}_a{c[1]}_c{c[2]}_d{c[3]}_n{c[4]}" for c in CASES])
def test_group_op_matches_the_oracle_and_the_per_layer_op(case, dtype, with_base):
    M, a, c, d, n = case
    gen = torch.Generator().manual_seed(M + a + c + d + n)
    x, x64 = rnd((M, a * d), dtype, gen)
    alphas = [0.7, 1.0, 0.4, 1.3][:n]
    w1s, w2s, bases, gs, f64 = [], [], [], [], []
    for i in range(n):
        w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
        w2, w2_64 = rnd((c, d), torch.float32, gen, 0.1)
        b, b64 = rnd((M, a * c), dtype, gen)
        g, g64 = rnd((M, a * c), dtype, gen, 0.1)
        w1s.append(torch.nn.Parameter(w1)); w2s.append(torch.nn.Parameter(w2)); bases.append(b); gs.append(g)
        f64.append((w1_64, w2_64, b64, g64))
    params = [p for pair in zip(w1s, w2s) for p in pair]

    def per_layer():
        xr = x.clone().requires_grad_(True)
        ys = [ops.lokr_linear(xr, w1s[i], w2s[i], alphas[i], bases[i] if with_base else None) for i in range(n)]
        grads = torch.autograd.grad(ys, [xr] + params, gs)
        return ys, grads

    def grouped():
        xr = x.clone().requires_grad_(True)
        ys = ops.lokr_linear_group(xr, w1s, w2s, alphas, bases if with_base else None)
        grads = torch.autograd.grad(ys, [xr] + params, gs)
        return ys, grads

    ys_p, gr_p = per_layer()
    ys_g, gr_g = grouped()
    torch.cuda.synchronize()
    errs, bounds = {}, {}
    dx_want = 0.0
    for i in range(n):
        assert torch.equal(ys_g[i], ys_p[i]), f"y[{i}] differs from the per-layer launch"
        w1_64, w2_64, b64, g64 = f64[i]
        y64 = oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=alphas[i])
        if with_base:  # fused epilogue: fp32 add, one rounding
            y64 = y64 + b64
        errs[f"y{i}"], bounds[f"y{i}"] = err(ys_g[i], y64, dtype), TOL["store_out"][dtype]
        gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=alphas[i])
        dx_want = dx_want + gr["dx"]
        errs[f"dw1_{i}"], bounds[f"dw1_{i}"] = err(gr_g[1 + 2 * i], gr["w1"]), TOL["f32_out"][dtype]
        errs[f"dw2_{i}"], bounds[f"dw2_{i}"] = err(gr_g[2 + 2 * i], gr["w2"]), TOL["f32_out"][dtype]
        assert torch.equal(gr_g[1 + 2 * i], gr_p[1 + 2 * i]) or err(gr_g[1 + 2 * i], gr_p[1 + 2 * i].double().cpu().numpy()) < 1e-6
    # the shared input's gradient.  No fused accumulation here, so the node takes its problem-by-problem path and the n 16-bit
    # results are added in 16 bits like autograd's accumulation: n roundings (the one-pass fp32 sum is the training-configuration test)
    errs["dx"], bounds["dx"] = err(gr_g[0], dx_want), (n + 1) * TOL["store_out"][dtype]
    errs["dx_vs_per_layer"] = err(gr_g[0], gr_p[0].double().cpu().numpy())
    bounds["dx_vs_per_layer"] = n * TOL["store_out"][dtype]
    check(f"sibling_group_op[{case},{dtype},{with_base}]", errs, bounds)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_sum_rows_is_the_fp32_sum_rounded_once(n, dtype):
    """lyc_sum_rows: dst = src[0] + ... + src[n - 1] with fp32 accumulation and one rounding (bit-exact against torch's fp32 sum),
    also in place (dst = src[0]) and on a size that is not a multiple of the grid"""
    import ctypes
    from lycoris_amd import _native as N
    gen = torch.Generator().manual_seed(n)
    for numel in (8, 1024 * 1280, 77 * 2048 + 8):
        srcs = [(torch.randn(numel, generator=gen) * 3).to(dtype).to(DEV) for _ in range(n)]
        want = torch.stack([s.float() for s in srcs]).sum(0).to(dtype) if n > 1 else srcs[0].clone()
        if n == 3:  # fp32 addition is not associative: the kernel adds in source order
            want = ((srcs[0].float() + srcs[1].float()) + srcs[2].float()).to(dtype)
        if n == 4:
            want = (((srcs[0].float() + srcs[1].float()) + srcs[2].float()) + srcs[3].float()).to(dtype)
        dst = torch.empty_like(srcs[0])
        arr = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
        N.call("lyc_sum_rows", ctypes.cast(arr, ctypes.c_void_p), n, N.ptr(dst), numel, N.dtype_code(dtype), N.stream_ptr(DEV))
        torch.cuda.synchronize()
        assert torch.equal(dst, want), (n, numel)
        N.call("lyc_sum_rows", ctypes.cast(arr, ctypes.c_void_p), n, N.ptr(srcs[0]), numel, N.dtype_code(dtype), N.stream_ptr(DEV))
        torch.cuda.synchronize()
        assert torch.equal(srcs[0], want), ("in place", n, numel)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", CASES + [(333, 8, 160, 160, 2)], ids=lambda c: f"M{c[0]}_a{c[1]}_c{c[2]}_d{c[3]}_n{c[4]}")
def test_group_op_in_the_training_configuration_accumulates_into_grad_and_reports_once(case, dtype):
    """fused accumulation + deferred, grouped weight gradients (the configuration bench.py and AdapterGradSync run): ONE backward launch
    for the set that stores the SUM of the n dx results (kron4_sum_kernel), .grad complete when backward() returns, one report per
    parameter"""
    M, a, c, d, n = case
    gen = torch.Generator().manual_seed(11 + M)
    x, x64 = rnd((M, a * d), dtype, gen)
    w1s, w2s, gs, f64 = [], [], [], []
    for i in range(n):
        w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
        w2, w2_64 = rnd((c, d), torch.float32, gen, 0.1)
        g, g64 = rnd((M, a * c), dtype, gen, 0.1)
        w1s.append(torch.nn.Parameter(w1)); w2s.append(torch.nn.Parameter(w2)); gs.append(g); f64.append((w1_64, w2_64, g64))
    params = [p for pair in zip(w1s, w2s) for p in pair]
    for p in params:
        p.grad = torch.zeros_like(p)
    reports = []
    ops.fused_grad_accumulation(True, callback=lambda p: reports.append(id(p)))
    try:
        xr = x.clone().requires_grad_(True)
        ys = ops.lokr_linear_group(xr, w1s, w2s, [1.0] * n)
        torch.autograd.backward(ys, gs)
        torch.cuda.synchronize()
    finally:
        ops.fused_grad_accumulation(False, None)
    assert sorted(reports) == sorted(id(p) for p in params)
    errs, bounds = {}, {}
    dx_want = 0.0
    for i in range(n):
        gr = oracle.lokr.backward(x64, f64[i][2], w1=f64[i][0], w2=f64[i][1])
        dx_want = dx_want + gr["dx"]
        errs[f"dw1_{i}"], bounds[f"dw1_{i}"] = err(w1s[i].grad, gr["w1"]), TOL["f32_out"][dtype]
        errs[f"dw2_{i}"], bounds[f"dw2_{i}"] = err(w2s[i].grad, gr["w2"]), TOL["f32_out"][dtype]
    # kron4_sum_kernel adds the n stage-2 results in fp32 registers and stores the sum ONCE: against the float64 sum rounded to the
    # storage type this is the north-star bound of a single layer's dx (n separate nodes: n roundings + n - 1 more in autograd's adds)
    errs["dx"], bounds["dx"] = err(xr.grad, dx_want, dtype), TOL["store_out"][dtype]
    check(f"sibling_group_op_training_configuration[{case},{dtype}]", errs, bounds)


@pytest.mark.parametrize("training", [False, True], ids=["autograd", "training_configuration"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", [(1024, 8, 160, 160, 3, 16), (77, 8, 160, 256, 2, 16), (130, 4, 24, 40, 4, 4)],
                         ids=lambda c: f"M{c[0]}_a{c[1]}_c{c[2]}_d{c[3]}_n{c[4]}_r{c[5]}")
def test_low_rank_group_op_matches_the_oracle(case, dtype, training):
    """lokr_linear_lr_group: the sibling set with w2 = w2a @ w2b (reference modules/lokr.py:131-136; BASELINE configs[3] "(low)"): planes
    packed from the pairs, one forward launch, one backward launch that stores the sum of the dx results, the weight gradients through
    the grouped chain rule -- forward bits equal to n lokr_linear_lr calls, every gradient against the float64 oracle"""
    M, a, c, d, n, r = case
    gen = torch.Generator().manual_seed(M + r + n)
    x, x64 = rnd((M, a * d), dtype, gen)
    w1s, w2as, w2bs, gs, f64 = [], [], [], [], []
    for i in range(n):
        w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
        w2a, a64 = rnd((c, r), torch.float32, gen, 0.3)
        w2b, b64 = rnd((r, d), torch.float32, gen, 0.3)
        g, g64 = rnd((M, a * c), dtype, gen, 0.1)
        w1s.append(torch.nn.Parameter(w1)); w2as.append(torch.nn.Parameter(w2a)); w2bs.append(torch.nn.Parameter(w2b)); gs.append(g)
        f64.append((w1_64, a64, b64, g64))
    params = [p for tri in zip(w1s, w2as, w2bs) for p in tri]
    alphas = [0.5, 1.0, 0.25, 0.75][:n]
    xp = x.clone().requires_grad_(True)
    ys_p = [ops.lokr_linear_lr(xp, w1s[i], w2as[i], w2bs[i], alphas[i]) for i in range(n)]
    if training:
        for p in params:
            p.grad = torch.zeros_like(p)
        ops.fused_grad_accumulation(True, callback=lambda p: None)
    try:
        xr = x.clone().requires_grad_(True)
        ys = ops.lokr_linear_lr_group(xr, w1s, w2as, w2bs, alphas)
        if training:
            torch.autograd.backward(ys, gs)
            grads = [xr.grad] + [p.grad for p in params]
        else:
            grads = list(torch.autograd.grad(ys, [xr] + params, gs))
        torch.cuda.synchronize()
    finally:
        if training:
            ops.fused_grad_accumulation(False, None)
    errs, bounds = {}, {}
    dx_want = 0.0
    for i in range(n):
        assert torch.equal(ys[i], ys_p[i]), f"y[{i}] differs from the per-layer op"
        w1_64, a64, b64, g64 = f64[i]
        gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2a=a64, w2b=b64, scale=alphas[i])
        dx_want = dx_want + gr["dx"]
        for j, k in enumerate(("w1", "w2a", "w2b")):
            errs[f"d{k}_{i}"], bounds[f"d{k}_{i}"] = err(grads[1 + 3 * i + j], gr[k]), TOL["f32_out"][dtype]
    if training:  # one launch, the sum formed in fp32 registers and rounded once
        errs["dx"], bounds["dx"] = err(grads[0], dx_want, dtype), TOL["store_out"][dtype]
    else:         # problem by problem, the n 16-bit results added like autograd's accumulation
        errs["dx"], bounds["dx"] = err(grads[0], dx_want), (n + 1) * TOL["store_out"][dtype]
    check(f"sibling_group_lr[{case},{dtype},{training}]", errs, bounds)


@pytest.mark.parametrize("training", [False, True], ids=["autograd", "training_configuration"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", [(1024, 1280, 1280, 16, 3), (77, 2048, 1280, 16, 2), (4096, 640, 640, 16, 3), (16384, 320, 320, 16, 3),
                                  (130, 96, 40, 4, 4), (256, 640, 640, 32, 2)], ids=lambda c: f"M{c[0]}_I{c[1]}_O{c[2]}_r{c[3]}_n{c[4]}")
def test_locon_group_op_matches_the_oracle_and_the_per_layer_op(case, dtype, training):
    """locon_linear_group: n LoCon projections of one input as one forward launch / one backward dx launch (bneck_group_kernel) -- forward
    bits equal to n locon_linear calls, every gradient against the float64 oracle; in the training configuration the factor gradients go
    through the grouped launches into .grad"""
    M, I, O, r, n = case
    gen = torch.Generator().manual_seed(M + I + O + r + n)
    x, x64 = rnd((M, I), dtype, gen)
    downs, ups, gs, f64 = [], [], [], []
    for i in range(n):
        dn, d64 = rnd((r, I), torch.float32, gen, 0.05)
        up, u64 = rnd((O, r), torch.float32, gen, 0.05)
        g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
        downs.append(torch.nn.Parameter(dn)); ups.append(torch.nn.Parameter(up)); gs.append(g); f64.append((d64, u64, g64))
    params = [p for pair in zip(downs, ups) for p in pair]
    alphas = [0.5, 1.0, 2.0, 0.25][:n]
    xp = x.clone().requires_grad_(True)
    ys_p = [ops.locon_linear(xp, downs[i], ups[i], alphas[i]) for i in range(n)]
    if training:
        for p in params:
            p.grad = torch.zeros_like(p)
        ops.fused_grad_accumulation(True, callback=lambda p: None)
    try:
        xr = x.clone().requires_grad_(True)
        ys = ops.locon_linear_group(xr, downs, ups, alphas)
        if training:
            torch.autograd.backward(ys, gs)
            grads = [xr.grad] + [p.grad for p in params]
        else:
            grads = list(torch.autograd.grad(ys, [xr] + params, gs))
        torch.cuda.synchronize()
    finally:
        if training:
            ops.fused_grad_accumulation(False, None)
    errs, bounds = {}, {}
    dx_want = 0.0
    for i in range(n):
        assert torch.equal(ys[i], ys_p[i]), f"y[{i}] differs from the per-layer op"
        d64, u64, g64 = f64[i]
        errs[f"y{i}"], bounds[f"y{i}"] = err(ys[i], oracle.locon.forward(x64, d64, u64, alphas[i], None), dtype), TOL["store_out"][dtype]
        dx_ref, dd_ref, du_ref = oracle.locon.backward(x64, g64, d64, u64, alphas[i], None)
        dx_want = dx_want + dx_ref
        errs[f"d_down{i}"], bounds[f"d_down{i}"] = err(grads[1 + 2 * i], dd_ref), TOL["f32_out"][dtype]
        errs[f"d_up{i}"], bounds[f"d_up{i}"] = err(grads[2 + 2 * i], du_ref), TOL["f32_out"][dtype]
    # n 16-bit results, summed in fp32 with one more rounding (training configuration) or added like autograd's accumulation
    errs["dx"], bounds["dx"] = err(grads[0], dx_want), (3 if training else n + 1) * TOL["store_out"][dtype]
    check(f"sibling_group_locon[{case},{dtype},{training}]", errs, bounds)


class Attn(nn.Module):
    """diffusers' Attention call pattern: to_q(h), to_k(ctx), to_v(ctx), to_out(.)"""

    def __init__(self, d, dc, heads=20):
        super().__init__()
        self.to_q, self.to_k, self.to_v, self.to_out = nn.Linear(d, d, bias=False), nn.Linear(dc, d, bias=False), nn.Linear(dc, d, bias=False), nn.Linear(d, d)
        self.heads = heads

    def forward(self, h, ctx=None):
        ctx = h if ctx is None else ctx
        q, k, v = self.to_q(h), self.to_k(ctx), self.to_v(ctx)
        B, T, D = q.shape
        sp = lambda t: t.view(B, -1, self.heads, D // self.heads).transpose(1, 2)
        a = torch.nn.functional.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, T, D)
        return self.to_out(a)


class Block(nn.Module):
    def __init__(self, d=1280, dc=2048):
        super().__init__()
        self.attn1, self.attn2 = Attn(d, d), Attn(d, dc)

    def forward(self, h, ctx):
        h = h + self.attn1(h)
        return h + self.attn2(h, ctx)


@pytest.mark.parametrize("rank", [10000, 16, -16], ids=["lokr_full_matrix", "lokr_rank16", "locon_rank16"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_adapted_attention_block_groups_its_projections_with_identical_results(dtype, rank):
    from lycoris_amd.modules import LoConModule
    torch.manual_seed(0)
    block = Block().to(DEV, dtype).requires_grad_(False)
    gen = torch.Generator().manual_seed(1)
    mods = []
    for name, layer in block.named_modules():
        if isinstance(layer, nn.Linear):
            if rank < 0:
                m = LoConModule(name.replace(".", "_"), layer, 1.0, -rank, 1).to(DEV)
                tgt = m.lora_up.weight
            else:
                m = LokrModule(name.replace(".", "_"), layer, 1.0, rank, 1, factor=8).to(DEV)
                tgt = m.lokr_w2 if m.use_w2 else m.lokr_w2_b
                assert m.use_w2 == (rank == 10000)
            with torch.no_grad():  # (the zero-initialised factor would make every delta zero)
                tgt.copy_((torch.randn(tgt.shape, generator=gen) * 0.05).to(DEV))
            m.apply_to()
            mods.append(m)
    params = [p for m in mods for p in m.parameters()]
    h0 = (torch.randn(1, 1024, 1280, generator=gen) * 0.5).to(DEV, dtype)
    ctx = (torch.randn(1, 77, 2048, generator=gen) * 0.5).to(DEV, dtype)
    gout = (torch.randn(1, 1024, 1280, generator=gen) * 0.05).to(DEV, dtype)

    def run():
        h = h0.clone().requires_grad_(True)
        y = block(h, ctx)
        grads = torch.autograd.grad(y, [h] + params, gout)
        torch.cuda.synchronize()
        return y.detach().clone(), [g.clone() for g in grads]

    for k in ("sets", "launches", "hits", "dissolved"):
        siblings._STATE[k] = 0
    siblings.enable(False)
    want = run()
    siblings.enable(True)
    learn = run()
    by = {m.lora_name: m for m in mods}
    assert [r().lora_name for r in by["attn1_to_q"]._sib.members] == ["attn1_to_q", "attn1_to_k", "attn1_to_v"]
    assert [r().lora_name for r in by["attn2_to_k"]._sib.members] == ["attn2_to_k", "attn2_to_v"]
    got = run()
    st = siblings.stats()
    assert st["launches"] == 2 and st["hits"] == 3 and st["dissolved"] == 0, st
    assert torch.equal(learn[0], want[0])
    # the forward pass is the same bits (same kernels, same accumulation order); the gradient of the shared input is summed in one
    # more order (n 16-bit results added in place), every other gradient comes from the same kernels
    assert torch.equal(got[0], want[0])
    errs = {"dh": float((got[1][0].float() - want[1][0].float()).norm() / want[1][0].float().norm())}
    bounds = {"dh": 3 * TOL["store_out"][dtype]}
    for i, (a_, b_) in enumerate(zip(got[1][1:], want[1][1:])):
        errs[f"g{i}"] = float((a_ - b_).norm() / (b_.norm() + 1e-30))
        bounds[f"g{i}"] = 2e-3  # downstream of dh's summation order (16-bit activations), not of the adapter kernels
    check(f"sibling_modules[{dtype},{rank}]", errs, bounds)
    for m in mods:
        m.restore()


class NormedAttn(nn.Module):
    """diffusers' BasicTransformerBlock in small: attn(norm(h)) -- under torch.autocast the LayerNorm's output is fp32"""

    def __init__(self, d=1280):
        super().__init__()
        self.norm, self.attn = nn.LayerNorm(d), Attn(d, d)

    def forward(self, h):
        return h + self.attn(self.norm(h))


@pytest.mark.parametrize("rank", [10000, -16], ids=["lokr_full_matrix", "locon_rank16"])
@pytest.mark.parametrize("layer_dtype", [torch.float32, torch.bfloat16], ids=["layers_f32", "layers_bf16"])
def test_sibling_sets_under_autocast_take_the_fp32_output_of_a_layernorm(layer_dtype, rank):
    """sd-scripts mixed precision (tests/test_gpu_autocast.py for the single layer): fp32 adapter parameters, the frozen block under
    torch.autocast(bf16), to_q / to_k / to_v called with the fp32 output of a LayerNorm.  The set forms on that fp32 tensor, the
    grouped op casts it once (the per-layer ops: once per projection); outputs are the same bits as the per-layer path."""
    from lycoris_amd.modules import LoConModule
    torch.manual_seed(0)
    block = NormedAttn().to(DEV, layer_dtype).requires_grad_(False)
    gen = torch.Generator().manual_seed(2)
    mods = []
    for name, layer in block.named_modules():
        if isinstance(layer, nn.Linear):
            if rank < 0:
                m = LoConModule(name.replace(".", "_"), layer, 1.0, -rank, 1).to(DEV)
                tgt = m.lora_up.weight
            else:
                m = LokrModule(name.replace(".", "_"), layer, 1.0, rank, 1, factor=8).to(DEV)
                tgt = m.lokr_w2
            with torch.no_grad():
                tgt.copy_((torch.randn(tgt.shape, generator=gen) * 0.05).to(DEV))
            m.apply_to()
            mods.append(m)
    params = [p for m in mods for p in m.parameters()]
    h0 = (torch.randn(1, 512, 1280, generator=gen) * 0.5).to(DEV)            # fp32 activations, as out of the fp32 residual stream
    gout = (torch.randn(1, 512, 1280, generator=gen) * 0.05).to(DEV)

    def run():
        h = h0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = block(h)
        grads = torch.autograd.grad(y, [h] + params, gout)
        torch.cuda.synchronize()
        return y.detach().clone(), [g.clone() for g in grads]

    for k in ("sets", "launches", "hits", "dissolved"):
        siblings._STATE[k] = 0
    siblings.enable(False)
    want = run()
    try:
        siblings.enable(True, autocast=True)
        run()  # learning pass
        q = {m.lora_name: m for m in mods}["attn_to_q"]
        assert q._sib is not None and [r().lora_name for r in q._sib.members] == ["attn_to_q", "attn_to_k", "attn_to_v"]
        got = run()
        st = siblings.stats()
        assert st["launches"] == 1 and st["hits"] == 2 and st["dissolved"] == 0, st
        assert got[0].dtype == want[0].dtype and torch.equal(got[0], want[0])
        errs = {"dh": float((got[1][0].float() - want[1][0].float()).norm() / want[1][0].float().norm())}
        bounds = {"dh": 3 * TOL["store_out"][torch.bfloat16]}
        for i, (a_, b_) in enumerate(zip(got[1][1:], want[1][1:])):
            errs[f"g{i}"] = float((a_ - b_).norm() / (b_.norm() + 1e-30))
            bounds[f"g{i}"] = 2e-3
        check(f"sibling_modules_autocast[{layer_dtype},{rank}]", errs, bounds)
        # outside autocast the same fp32 tensor is not a 16-bit activation: per-layer path, the set is kept for the next autocast pass
        h = h0.clone()
        y32 = block(h) if layer_dtype == torch.float32 else None
        assert siblings.stats()["launches"] == 1 and (y32 is None or y32.dtype == torch.float32)
        for m in mods:
            m.restore()
    finally:
        siblings.enable(True, autocast=False)
