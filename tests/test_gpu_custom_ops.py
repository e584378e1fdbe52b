"""GPU: the TORCH_LIBRARY(lycoris_amd) custom-op path (csrc/torch_ops.cpp, the default dispatch of lycoris_amd.ops).

Every other -m gpu test already goes through it; here: (a) it agrees bit for bit with the ctypes / Python autograd.Function
statement of the same host logic (same kernels, same launch parameters), (b) fused gradient accumulation into .grad and the
grad-sync callback work from the C++ backward, (c) an adapted layer traces under torch.compile (FakeTensor kernels,
backward split into dispatcher ops) -- the reference exercises torch.compile in test/compile.py."""
import pytest
import torch
import torch._dynamo
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both(fn):
    from lycoris_amd import ops
    out = {}
    for mode in ("cpp", "python"):
        ops.set_dispatch(mode)
        try:
            out[mode] = fn()
        finally:
            ops.set_dispatch("cpp")
    return out["cpp"], out["python"]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("algo", ["lokr", "locon", "loha", "chan", "lokr_conv", "locon_conv"])
def test_cpp_dispatch_equals_python_dispatch(algo, dtype):
    from lycoris_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    rn = lambda *s, sc=1.0, dt=torch.float32: (torch.randn(*s, device=DEV, generator=g) * sc).to(dt).requires_grad_(True)
    if algo in ("lokr_conv", "locon_conv"):
        if dtype == torch.float32:
            pytest.skip("the implicit Conv2d kernels take 16-bit activations")
        x = rn(2, 64, 9, 8, dt=dtype)
        fs = [rn(8, 8, sc=0.3), rn(16, 8, 3, 3, sc=0.1)] if algo == "lokr_conv" else [rn(8, 64, 3, 3, sc=0.1), rn(48, 8, 1, 1, sc=0.1)]
        f = (lambda: ops.lokr_conv2d(x, *fs, 0.7, (1, 1), (1, 1), (1, 1))) if algo == "lokr_conv" else \
            (lambda: ops.locon_conv2d(x, *fs, 0.7, (1, 1), (1, 1), (1, 1)))
    else:
        x = rn(3, 50, 64, dt=dtype)
        if algo == "lokr":
            fs = [rn(8, 8, sc=0.3), rn(16, 8, sc=0.1)]
            f = lambda: ops.lokr_linear(x, *fs, 0.7)
        elif algo == "locon":
            fs = [rn(8, 64, sc=0.1), rn(40, 8, sc=0.1)]
            f = lambda: ops.locon_linear(x, *fs, 0.7)
        elif algo == "loha":
            fs = [rn(40, 4, sc=0.3), rn(4, 64), rn(40, 4, sc=0.3), rn(4, 64)]
            f = lambda: ops.loha_linear(x, *fs, 0.7)
        else:
            fs = [rn(64, sc=0.3)]
            bias = torch.randn(64, device=DEV)
            f = lambda: ops.chan_affine(x, fs[0], bias, 1.0, 0.7, -1)

    def run():
        y = f()
        gy = torch.ones_like(y) * 0.01
        return [y.detach().clone()] + [t.clone() for t in torch.autograd.grad(y, [x] + fs, gy)]

    a, b = _both(run)
    for i, (u, v) in enumerate(zip(a, b)):
        assert u.dtype == v.dtype and u.shape == v.shape, i
        # atomics make the factor gradients order-dependent in the last bits; y / dx are deterministic
        if i < 2:
            assert torch.equal(u, v), (algo, i, float((u.float() - v.float()).abs().max()))
        else:
            assert torch.allclose(u.float(), v.float(), rtol=1e-4, atol=1e-6), (algo, i)


def test_fused_accumulation_and_callback_from_the_cpp_backward():
    from lycoris_amd import ops
    x = torch.randn(64, 64, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    w1 = nn.Parameter(torch.randn(8, 8, device=DEV) * 0.3)
    w2 = nn.Parameter(torch.randn(16, 8, device=DEV) * 0.1)
    y = ops.lokr_linear(x, w1, w2, 1.0)
    g = torch.randn_like(y) * 0.1
    want = torch.autograd.grad(y, [x, w1, w2], g)          # plain autograd: gradients handed back
    seen = []
    w1.grad, w2.grad = torch.zeros_like(w1), torch.zeros_like(w2)
    ops.fused_grad_accumulation(True, callback=lambda p: seen.append(p))
    try:
        y = ops.lokr_linear(x, w1, w2, 1.0)
        # the factors are asked for: the kernels add their gradients into `.grad` and hand nothing back (hence allow_unused)
        dx, h1, h2 = torch.autograd.grad(y, [x, w1, w2], g, allow_unused=True)
        assert h1 is None and h2 is None
        assert seen and {id(p) for p in seen} == {id(w1), id(w2)}     # both parameters reported, from C++
        assert torch.equal(dx, want[0])
        assert torch.allclose(w1.grad, want[1], rtol=1e-4, atol=1e-6) and torch.allclose(w2.grad, want[2], rtol=1e-4, atol=1e-6)
        y = ops.lokr_linear(x, w1, w2, 1.0)
        y.backward(g)                                                      # second micro-batch through .backward(): += on top
        assert torch.allclose(w2.grad, 2 * want[2], rtol=1e-4, atol=1e-6)
    finally:
        ops.fused_grad_accumulation(False, None)


@pytest.mark.parametrize("algo", ["lokr", "locon", "loha", "ia3"])
def test_a_gradient_probe_leaves_the_factor_gradients_alone(algo):
    """torch.autograd.grad(y, [x]) asks for dx only: with fused accumulation on, the factors' `.grad` must not change and nothing
    is reported to the DP bucket counters (ADVICE r2 / VERDICT r3 #8: needs_input_grad is honoured, csrc/torch_ops.cpp)"""
    from lycoris_amd import ops
    x = torch.randn(64, 64, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    rn = lambda *s, sc=0.2: nn.Parameter(torch.randn(*s, device=DEV) * sc)
    if algo == "lokr":
        fs = [rn(8, 8), rn(16, 8)]
        f = lambda: ops.lokr_linear(x, fs[0], fs[1], 1.0)
    elif algo == "locon":
        fs = [rn(8, 64), rn(96, 8)]
        f = lambda: ops.locon_linear(x, fs[0], fs[1], 1.0)
    elif algo == "loha":
        fs = [rn(96, 4), rn(4, 64), rn(96, 4), rn(4, 64)]
        f = lambda: ops.loha_linear(x, *fs, 1.0)
    else:
        fs = [rn(64, sc=1.0)]
        f = lambda: ops.chan_affine(x, fs[0], None, 0.0, 1.0, -1)
    seen = []
    for p in fs:
        p.grad = torch.full_like(p, 3.0)
    ops.fused_grad_accumulation(True, callback=lambda p: seen.append(p))
    try:
        y = f()
        dx, = torch.autograd.grad(y, [x], torch.randn_like(y))
        torch.cuda.synchronize()
        assert dx is not None and not seen
        for p in fs:
            assert torch.equal(p.grad, torch.full_like(p, 3.0))
    finally:
        ops.fused_grad_accumulation(False, None)


ROWS_CONV = [("loha", 3, 1, 1), ("loha", 1, 1, 0), ("lokr", 3, 2, 1), ("lokr", 1, 1, 0), ("locon", 3, 1, 1), ("locon", 1, 1, 0)]


@pytest.mark.parametrize("channels_last", [False, True], ids=["nchw", "nhwc"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("algo,k,stride,pad", ROWS_CONV, ids=[f"{a}_k{k}_s{s}" for a, k, s, _ in ROWS_CONV])
def test_rows_lowered_conv2d_cpp_dispatch_equals_python_dispatch(algo, k, stride, pad, dtype, channels_last):
    """torch.ops.lycoris_amd.adapter_conv2d (round 3): the im2col / NHWC-rows lowering of every adapter as ONE C++ op with the
    same C ABI calls as the Python autograd.Function it replaces -- LoHa on any Conv2d, LoKr / LoCon off their implicit
    kernels (fp32 activations; a 3-wide w1 here for the 16-bit case), 1x1 convolutions"""
    from lycoris_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    rn = lambda *s, sc=1.0, dt=torch.float32: (torch.randn(*s, device=DEV, generator=g) * sc).to(dt).requires_grad_(True)
    C, O = 48, 72
    x = rn(2, C, 9, 8, dt=dtype)
    if channels_last:
        x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    geo = ((stride, stride), (pad, pad), (1, 1))
    if algo == "loha":
        fs = [rn(O, 4, sc=0.3), rn(4, C * k * k), rn(O, 4, sc=0.3), rn(4, C * k * k)]
        f = lambda: ops.loha_conv2d(x, *fs, 0.7, (O, C, k, k), *geo)
    elif algo == "lokr":  # w1 3 x 3: not a shape of the implicit kernel
        fs = [rn(3, 3, sc=0.3), rn(O // 3, C // 3, k, k, sc=0.1)]
        f = lambda: ops.lokr_conv2d(x, *fs, 0.7, *geo)
    else:  # rank 6: off the implicit LoCon kernel's rank tiles for 16-bit; fp32 always lowers
        fs = [rn(6, C, k, k, sc=0.1), rn(O, 6, 1, 1, sc=0.1)]
        f = lambda: ops.locon_conv2d(x, *fs, 0.7, *geo)

    def run():
        y = f()
        gy = torch.ones_like(y) * 0.01
        return [y.detach().clone()] + [t.clone() for t in torch.autograd.grad(y, [x] + fs, gy)]

    a, b = _both(run)
    for i, (u, v) in enumerate(zip(a, b)):
        assert u.dtype == v.dtype and u.shape == v.shape, i
        if i < 2:
            if dtype == torch.float32:  # the fp32 row kernels split K with atomics: equal to the last bits only
                assert torch.allclose(u, v, rtol=1e-5, atol=1e-6), (algo, i, float((u - v).abs().max()))
            else:
                assert torch.equal(u, v), (algo, i, float((u.float() - v.float()).abs().max()))
            assert u.is_contiguous(memory_format=torch.channels_last) == v.is_contiguous(memory_format=torch.channels_last), i
        else:
            assert torch.allclose(u.float(), v.float(), rtol=1e-4, atol=1e-6), (algo, i)


@pytest.mark.parametrize("algo", ["lokr", "locon", "loha"])
def test_adapted_linear_layer_under_torch_compile(algo):
    """torch.compile(backend="aot_eager"): Dynamo + AOTAutograd trace forward AND backward of the adapted layer through the
    FakeTensor (Meta) kernels of the custom ops -- no graph break, same numbers as eager.  (aot_eager: the surrounding ops
    are not handed to a code generator; what is checked is that OUR ops are traceable and differentiable.)"""
    from lycoris_amd.modules import LoConModule, LohaModule, LokrModule
    torch.manual_seed(0)
    layer = nn.Linear(64, 128).to(DEV, torch.bfloat16).requires_grad_(False)
    cls, kw = {"lokr": (LokrModule, dict(lora_dim=100000, alpha=1, factor=8)), "locon": (LoConModule, dict(lora_dim=8, alpha=4)),
               "loha": (LohaModule, dict(lora_dim=4, alpha=2))}[algo]
    mod = cls("m", layer, 1.0, **kw).to(DEV)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    mod.apply_to()
    x = torch.randn(5, 7, 64, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    params = list(mod.parameters())

    def run(fn):
        y = fn(x)
        return [y.detach()] + list(torch.autograd.grad(y.float().pow(2).sum(), [x] + params))

    eager = run(layer)
    torch._dynamo.reset()
    compiled = torch.compile(layer, backend="aot_eager", fullgraph=True)
    got = run(compiled)
    mod.restore()
    for i, (u, v) in enumerate(zip(got, eager)):
        if i == 1:
            # dx = g W + dx_adapter.  Eager LoKr holds the frozen layer in the adapter's node (round 6: g W joins the adapter's dx inside the
            # library GEMM's fp32 epilogue, one rounding); the traced graph adds two rounded tensors.  Where the two terms cancel the
            # results differ by an ulp of the LARGER term: compare norm-wise
            assert float((u.float() - v.float()).norm() / v.float().norm()) <= 1e-2, (algo, i)
            continue
        assert torch.allclose(u.float(), v.float(), rtol=2e-2 if i < 2 else 1e-3, atol=1e-3), (algo, i)


@pytest.mark.parametrize("ksize", [3, 1], ids=["k3", "k1"])
@pytest.mark.parametrize("algo", ["lokr", "locon", "loha"])
def test_adapted_conv2d_layer_under_torch_compile(algo, ksize):
    """the Conv2d ops trace too: forward = the public op, backward = lycoris_amd::_{lokr,locon}_conv2d_backward (implicit kernels)
    or lycoris_amd::_adapter_conv2d_backward (LoHa; every 1x1 convolution), all with Meta kernels (reference: test/compile.py
    compiles a model with conv layers)"""
    from lycoris_amd.modules import LoConModule, LohaModule, LokrModule
    torch.manual_seed(0)
    layer = nn.Conv2d(64, 128, ksize, padding=ksize // 2).to(DEV, torch.bfloat16).requires_grad_(False)
    cls, kw = {"lokr": (LokrModule, dict(lora_dim=100000, alpha=1, factor=8)), "locon": (LoConModule, dict(lora_dim=8, alpha=4)),
               "loha": (LohaModule, dict(lora_dim=4, alpha=2))}[algo]
    mod = cls("m", layer, 1.0, **kw).to(DEV)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    mod.apply_to()
    x = torch.randn(2, 64, 9, 8, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    params = list(mod.parameters())

    def run(fn):
        y = fn(x)
        return [y.detach()] + list(torch.autograd.grad(y.float().pow(2).sum(), [x] + params))

    eager = run(layer)
    torch._dynamo.reset()
    compiled = torch.compile(layer, backend="aot_eager", fullgraph=True)
    got = run(compiled)
    mod.restore()
    for i, (u, v) in enumerate(zip(got, eager)):
        assert u.shape == v.shape, (algo, i)
        assert torch.allclose(u.float(), v.float(), rtol=2e-2 if i < 2 else 1e-3, atol=2e-3), (algo, i, float((u.float() - v.float()).abs().max()))


class _ToyBlock(nn.Module):
    """conv -> 1x1 conv -> (tokens) linear -> linear -> linear, the frozen layers of a mixed-algorithm preset"""

    def __init__(self):
        super().__init__()
        self.conv_in = nn.Conv2d(32, 64, 3, padding=1)
        self.proj = nn.Conv2d(64, 64, 1)
        self.q = nn.Linear(64, 64)
        self.ff1 = nn.Linear(64, 128)
        self.ff2 = nn.Linear(128, 64)

    def forward(self, x):
        h = self.proj(torch.nn.functional.silu(self.conv_in(x)))
        t = h.flatten(2).transpose(1, 2)
        t = t + self.q(t)
        return t + self.ff2(torch.nn.functional.gelu(self.ff1(t)))


def test_mixed_algorithm_block_under_torch_compile_fullgraph():
    """one graph, no break, over a block that mixes every algorithm and both layer types --
    LoCon on the 3x3 conv, LoHa on the 1x1 conv, LoKr / LoHa / (IA)^3 on the Linears (reference: test/compile.py)"""
    from lycoris_amd.modules import IA3Module, LoConModule, LohaModule, LokrModule
    torch.manual_seed(1)
    block = _ToyBlock().to(DEV, torch.bfloat16).requires_grad_(False)
    plan = [("conv_in", LoConModule, dict(lora_dim=8, alpha=4)), ("proj", LohaModule, dict(lora_dim=4, alpha=2)),
            ("q", LokrModule, dict(lora_dim=100000, alpha=1, factor=8)), ("ff1", LohaModule, dict(lora_dim=4, alpha=2)),
            ("ff2", IA3Module, dict())]
    mods = []
    for name, cls, kw in plan:
        m = cls(name, getattr(block, name), 1.0, **kw).to(DEV)
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn_like(p) * 0.2)
        m.apply_to()
        mods.append(m)
    params = [p for m in mods for p in m.parameters()]
    x = torch.randn(2, 32, 8, 8, device=DEV, dtype=torch.bfloat16, requires_grad=True)

    def run(fn):
        y = fn(x)
        return [y.detach()] + list(torch.autograd.grad(y.float().pow(2).sum(), [x] + params))

    eager = run(block)
    torch._dynamo.reset()
    compiled = torch.compile(block, backend="aot_eager", fullgraph=True)
    got = run(compiled)
    for m in mods:
        m.restore()
    for i, (u, v) in enumerate(zip(got, eager)):
        assert u.shape == v.shape, i
        den = float(v.float().norm()) or 1.0
        assert float((u.float() - v.float()).norm()) / den <= 2e-2, (i, float((u.float() - v.float()).norm()) / den)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(64, 8, 32, 32), (300, 8, 160, 160), (77, 8, 160, 256), (33, 4, 20, 24), (1024, 8, 1280, 160)],
                         ids=lambda s: "x".join(map(str, s)))
def test_lokr_fused_base_plus_delta(shape, dtype):
    """y = base + delta in the kron3 epilogue (fp32 add, ONE rounding) vs the oracle; gradients: d base = g, the rest as
    without the fusion.  (33, 4, 20, 24): c % 4 == 0 but ragged tiles; the module-level use is LokrModule.forward."""
    import numpy as np
    import oracle
    from gpu_util import TOL, check, err, rnd
    from lycoris_amd import ops
    M, G, c, d = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x, x64 = rnd((M, G * d), dtype, gen)
    base, b64 = rnd((M, G * c), dtype, gen)
    g, g64 = rnd((M, G * c), dtype, gen, 1.0 / np.sqrt(G * c))
    w1, n1 = rnd((G, G), torch.float32, gen, 0.3)
    w2, n2 = rnd((c, d), torch.float32, gen, 0.1)
    for t in (x, base, w1, w2):
        t.requires_grad_(True)
    assert ops.lokr_linear_fusable(x, w1, w2, base)
    y = ops.lokr_linear(x, w1, w2, 0.75, base=base)
    dx, db, dw1, dw2 = torch.autograd.grad(y, [x, base, w1, w2], g)
    y_ref = b64 + oracle.lokr.forward(x64, w1=n1, w2=n2, scale=0.75)
    gr = oracle.lokr.backward(x64, g64, w1=n1, w2=n2, scale=0.75)
    assert torch.equal(db, g)
    # unfused: round(base + round(delta)); fused: round(base + delta) -- the fused value is the closer one to the oracle
    unfused = (base + ops.lokr_linear(x, w1, w2, 0.75)).detach()
    e_f, e_u = err(y, y_ref, dtype), err(unfused, y_ref, dtype)
    assert e_f <= e_u * 1.05 + 1e-6, (e_f, e_u)
    check(f"lokr_fused_base[{shape},{dtype}]",
          {"y": e_f, "dx": err(dx, gr["dx"], dtype), "dw1": err(dw1, gr["w1"]), "dw2": err(dw2, gr["w2"])},
          {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype], "dw1": TOL["f32_out"][dtype], "dw2": TOL["f32_out"][dtype]})


def test_lokr_module_forward_uses_the_fused_epilogue():
    from lycoris_amd import ops
    from lycoris_amd.modules import LokrModule
    torch.manual_seed(1)
    layer = nn.Linear(64, 128).to(DEV, torch.bfloat16).requires_grad_(False)
    mod = LokrModule("m", layer, 0.7, lora_dim=100000, alpha=1, factor=8).to(DEV)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    x = torch.randn(5, 7, 64, device=DEV, dtype=torch.bfloat16)
    base = layer(x)
    want = base.float() + mod.bypass_forward_diff(x, scale=mod.multiplier).float()
    calls, owned = [], []
    orig, orig_owned = ops.lokr_linear, ops.lokr_adapted_linear
    ops.lokr_linear = lambda *a, **k: (calls.append(k.get("base") is not None), orig(*a, **k))[1]
    ops.lokr_adapted_linear = lambda *a, **k: (owned.append(len(a[1])), orig_owned(*a, **k))[1]
    try:
        mod.apply_to()
        out_owned = layer(x)          # a frozen 16-bit nn.Linear: layer + adapter in ONE node (round 6), base + delta fused in it
        layer.weight.requires_grad_(True)
        out = layer(x)                # a trainable base weight: F.linear, then the adapter kernel with the fused epilogue
        layer.weight.requires_grad_(False)
        mod.restore()
    finally:
        ops.lokr_linear, ops.lokr_adapted_linear = orig, orig_owned
    assert owned == [1] and calls == [True]  # one call each; the second with base
    assert torch.equal(out_owned, out)
    assert torch.allclose(out.float(), want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("algo", ["locon", "loha"])
def test_pointwise_conv_on_a_channels_last_tensor_is_copy_free_and_equal(algo):
    """a 1x1 conv is the row op on the NHWC pixel rows: a channels_last activation is used as it is and the output keeps the
    memory format (reference: F.conv2d preserves channels_last, functional/general.py:6)"""
    from lycoris_amd import ops
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(2, 64, 6, 5, device=DEV, generator=g).to(torch.bfloat16)
    gy = (torch.randn(2, 48, 6, 5, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    if algo == "locon":
        fs = [torch.randn(8, 64, 1, 1, device=DEV, generator=g) * 0.1, torch.randn(48, 8, 1, 1, device=DEV, generator=g) * 0.1]
        f = lambda t: ops.locon_conv2d(t, *fs, 0.7, (1, 1), (0, 0), (1, 1))
    else:
        fs = [torch.randn(48, 4, device=DEV, generator=g) * 0.3, torch.randn(4, 64, device=DEV, generator=g),
              torch.randn(48, 4, device=DEV, generator=g) * 0.3, torch.randn(4, 64, device=DEV, generator=g)]
        f = lambda t: ops.loha_conv2d(t, *fs, 0.7, (48, 64, 1, 1), (1, 1), (0, 0), (1, 1))
    for t in fs:
        t.requires_grad_(True)
    outs = []
    for cl in (False, True):
        xi = (x.contiguous(memory_format=torch.channels_last) if cl else x.clone()).requires_grad_(True)
        gi = gy.contiguous(memory_format=torch.channels_last) if cl else gy
        y = f(xi)
        assert y.is_contiguous(memory_format=torch.channels_last) == cl
        grads = torch.autograd.grad(y, [xi] + fs, gi)
        assert grads[0].is_contiguous(memory_format=torch.channels_last) == cl
        outs.append([y.detach().float()] + [t.float() for t in grads])
    for u, v in zip(*outs):
        assert torch.allclose(u, v, rtol=1e-4, atol=1e-6), float((u - v).abs().max())


@pytest.mark.parametrize("with_bias", [False, True], ids=["in_side", "out_side"])
def test_chan_affine_on_a_channels_last_tensor_is_copy_free_and_equal(with_bias):
    """(IA)^3 on nn.Conv2d: scaling the channel dimension of a channels_last tensor is the last-dimension case on its NHWC view --
    same numbers as for the NCHW tensor, result in the caller's memory format (round 3: `.contiguous()` used to copy twice per pass)"""
    from lycoris_amd import ops
    g = torch.Generator(device=DEV).manual_seed(9)
    a = torch.randn(2, 48, 9, 8, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(1, 48, 1, 1, device=DEV, generator=g) * 0.3).requires_grad_(True)
    bias = torch.randn(48, device=DEV, generator=g) if with_bias else None
    gy = (torch.randn(2, 48, 9, 8, device=DEV, generator=g) * 0.1).to(torch.bfloat16)

    def run(cl):
        x = a.clone()
        go = gy.clone()
        if cl:
            x, go = x.contiguous(memory_format=torch.channels_last), go.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        y = ops.chan_affine(x, w, bias, 1.0 if with_bias else 0.0, 0.7, 1)
        dx, dw = torch.autograd.grad(y, [x, w], go)
        return y, dx, dw

    y0, dx0, dw0 = run(False)
    y1, dx1, dw1 = run(True)
    assert y1.is_contiguous(memory_format=torch.channels_last) and dx1.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y0, y1.contiguous()) and torch.equal(dx0, dx1.contiguous())
    assert torch.allclose(dw0, dw1, rtol=1e-4, atol=1e-6)
