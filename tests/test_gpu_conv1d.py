"""GPU: adapters on nn.Conv1d (modules/base.py _Conv1dTwin: the layer is adapted as a Conv2d with a 1 x k window).

(a) the adapted Conv1d layer computes exactly what the same adapter computes on the equivalent nn.Conv2d layer over [B, C, 1, L]
    (same adapter kernels, same parameters; the frozen op is MIOpen's conv1d there and conv2d here);
(b) against torch: layer(x) == F.conv1d(x, W + dW) with the module's own get_diff_weight, within the storage-rounding bound."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [("locon", dict(lora_dim=8, alpha=4)), ("locon_tucker", dict(lora_dim=4, alpha=2, use_tucker=True)),
         ("locon_dora", dict(lora_dim=8, alpha=4, weight_decompose=True)), ("loha", dict(lora_dim=4, alpha=2)),
         ("lokr", dict(lora_dim=100000, alpha=1, factor=8)), ("lokr_lowrank", dict(lora_dim=2, alpha=1, factor=8)), ("ia3", dict())]


def _cls(name):
    from lycoris_amd.modules import IA3Module, LoConModule, LohaModule, LokrModule
    return {"locon": LoConModule, "loha": LohaModule, "lokr": LokrModule, "ia3": IA3Module}[name.split("_")[0]]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_conv1d_adapter_equals_the_conv2d_adapter_on_the_lifted_layer(name, kw, dtype):
    torch.manual_seed(3)
    C, O, k, s, p, B, L = 64, 128, 3, 1, 1, 2, 40
    l1 = nn.Conv1d(C, O, k, s, p).to(DEV, dtype).requires_grad_(False)
    l2 = nn.Conv2d(C, O, (1, k), (1, s), (0, p)).to(DEV, dtype).requires_grad_(False)
    with torch.no_grad():
        l2.weight.copy_(l1.weight.unsqueeze(2))
        l2.bias.copy_(l1.bias)
    m1 = _cls(name)("a", l1, 1.0, **kw).to(DEV)
    m2 = _cls(name)("b", l2, 1.0, **kw).to(DEV)
    with torch.no_grad():
        for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert n1 == n2 and p1.shape == p2.shape, (n1, p1.shape, p2.shape)
            if n1 != "dora_scale":
                p1.copy_(torch.randn_like(p1) * 0.2)
            p2.copy_(p1)
    m1.apply_to()
    m2.apply_to()
    x = torch.randn(B, C, L, device=DEV, dtype=dtype, requires_grad=True)
    gy = torch.randn(B, O, L, device=DEV, dtype=dtype) * 0.1
    y1 = l1(x)
    g1 = torch.autograd.grad(y1, [x] + list(m1.parameters()), gy)
    x2 = x.detach().clone().requires_grad_(True)
    y2 = l2(x2.unsqueeze(2)).squeeze(2)
    g2 = torch.autograd.grad(y2, [x2] + list(m2.parameters()), gy)
    torch.cuda.synchronize()
    assert y1.shape == (B, O, L)
    # the FROZEN op differs (MIOpen's conv1d vs conv2d kernels), the adapter kernels are the same: equal up to the frozen op's rounding
    rel = lambda u, v: float((u.detach().float() - v.detach().float()).norm() / (v.detach().float().norm() + 1e-30))
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-5
    assert rel(y1, y2) <= tol, rel(y1, y2)
    for i, (u, v) in enumerate(zip(g1, g2)):
        assert u.shape == v.shape
        assert rel(u, v) <= (tol if i == 0 else max(tol, 2e-4)), (i, rel(u, v))
    m1.restore()
    m2.restore()
    # (b) against torch, through the module's own dW (not for (IA)^3 / DoRA: no plain dW)
    if name.split("_")[0] in ("locon", "loha", "lokr") and "dora" not in name:
        dw = m1.get_diff_weight()[0].detach()
        assert dw.shape == l1.weight.shape
        want = nn.functional.conv1d(x.detach().float(), l1.weight.float() + dw.float(), l1.bias.float(), s, p)
        tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
        assert float((y1.float() - want).norm() / want.norm()) <= tol


def test_conv1d_stride_and_dilation():
    """stride 2, dilation 2, no padding: the geometry goes through as (1, s) / (1, d) / (0, p)"""
    from lycoris_amd.modules import LoConModule
    torch.manual_seed(4)
    l1 = nn.Conv1d(32, 48, 5, stride=2, padding=0, dilation=2).to(DEV, torch.float32).requires_grad_(False)
    m = LoConModule("a", l1, 1.0, lora_dim=4, alpha=4).to(DEV)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    x = torch.randn(3, 32, 50, device=DEV)
    base = l1(x)
    m.apply_to()
    y = l1(x)
    m.restore()
    dw = m.get_diff_weight()[0]
    want = base + nn.functional.conv1d(x, dw, None, 2, 0, 2)
    assert y.shape == want.shape and float((y - want).norm() / want.norm()) <= 1e-4
