"""Sibling projections behind the module API (modules/siblings.py): the set-finding logic on host tensors (the GPU parity of the
grouped launches is tests/test_gpu_lokr_group.py).  The eligibility test of LokrModule asks for a 16-bit HIP tensor; here it is
relaxed to "any tensor" so that the same code runs on the CPU, where ops.lokr_linear_group evaluates the problems one by one."""
import pytest
import torch
import torch.nn as nn

from lycoris_amd.modules import LoConModule, LokrModule
from lycoris_amd.modules import siblings


class Attn(nn.Module):
    """the call pattern of diffusers' Attention: to_q(h), to_k(ctx), to_v(ctx), to_out(.)"""

    def __init__(self, d=64, dc=64):
        super().__init__()
        self.to_q, self.to_k, self.to_v, self.to_out = nn.Linear(d, d), nn.Linear(dc, d), nn.Linear(dc, d), nn.Linear(d, d)

    def forward(self, h, ctx=None):
        ctx = h if ctx is None else ctx
        q, k, v = self.to_q(h), self.to_k(ctx), self.to_v(ctx)
        a = torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v
        return self.to_out(a)


class Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.attn1, self.attn2 = Attn(64, 64), Attn(64, 32)

    def forward(self, h, ctx):
        h = h + self.attn1(h)
        return h + self.attn2(h, ctx)


@pytest.fixture()
def host_eligible(monkeypatch):
    monkeypatch.setattr(LokrModule, "_sibling_eligible",
                        lambda self, x: self.module_type == "linear" and self.use_w1 and (self.use_w2 or not self.tucker) and not self.wd
                        and not (self.training and (self.module_dropout or self.rank_dropout)))
    monkeypatch.setattr(LoConModule, "_sibling_eligible",
                        lambda self, x: not self.isconv and not getattr(self, "wd", False)
                        and not (self.training and (self.module_dropout or self.rank_dropout)))
    siblings.enable(True)
    for k in ("sets", "launches", "hits", "dissolved"):
        siblings._STATE[k] = 0
    yield
    siblings.enable(True)


def adapt(block, seed=0):
    g = torch.Generator().manual_seed(seed)
    mods = []
    for name, layer in block.named_modules():
        if isinstance(layer, nn.Linear):
            m = LokrModule(name.replace(".", "_"), layer, 1.0, 10000, 1, factor=4)
            with torch.no_grad():
                m.lokr_w2.copy_(torch.randn(m.lokr_w2.shape, generator=g) * 0.1)
            m.apply_to()
            mods.append(m)
    return mods


def run(block, h, ctx):
    h = h.clone().requires_grad_(True)
    y = block(h, ctx)
    params = [p for m in block._mods for p in m.parameters()]
    grads = torch.autograd.grad(y.square().sum(), [h] + params)
    return y.detach(), [g.clone() for g in grads]


def test_sets_form_from_the_call_pattern_and_change_nothing_but_the_launch_count(host_eligible):
    torch.manual_seed(0)
    block = Block()
    block._mods = adapt(block)
    h, ctx = torch.randn(2, 9, 64), torch.randn(2, 5, 32)
    siblings.enable(False)
    want = run(block, h, ctx)
    siblings.enable(True)
    first = run(block, h, ctx)            # learning pass: per-layer path, sets are formed
    by_name = {m.lora_name: m for m in block._mods}
    s1, s2 = by_name["attn1_to_q"]._sib, by_name["attn2_to_k"]._sib
    assert s1 is not None and [r().lora_name for r in s1.members] == ["attn1_to_q", "attn1_to_k", "attn1_to_v"]
    assert s2 is not None and [r().lora_name for r in s2.members] == ["attn2_to_k", "attn2_to_v"]
    assert by_name["attn2_to_q"]._sib is None and by_name["attn1_to_out"]._sib is None
    assert siblings.stats()["sets"] == 2 and siblings.stats()["launches"] == 0
    second = run(block, h, ctx)           # steady state: 2 grouped launches, 3 parked results fetched
    st = siblings.stats()
    assert st["launches"] == 2 and st["hits"] == 3 and st["dissolved"] == 0
    for got in (first, second):
        assert torch.allclose(got[0], want[0], atol=1e-6)
        for a, b in zip(got[1], want[1]):
            assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)
    third = run(block, torch.randn(2, 9, 64), ctx)  # new tensors, same pattern
    assert siblings.stats()["launches"] == 4 and siblings.stats()["dissolved"] == 0 and third[0].shape == want[0].shape


def test_a_changed_call_pattern_dissolves_the_set_and_the_result_stays_right(host_eligible):
    torch.manual_seed(1)
    attn = Attn()
    attn._mods = adapt(attn)
    h = torch.randn(3, 7, 64)
    attn(h)  # learn: q, k, v share h
    q = attn._mods[0]
    assert q._sib is not None and len(q._sib.members) == 3
    other = torch.randn(3, 7, 64)
    siblings.enable(False)
    want_k = attn.to_k(other)
    siblings.enable(True)
    yq = attn.to_q(h)                 # leader: runs the group, parks k and v
    assert len(q._sib.pending) == 2
    got_k = attn.to_k(other)          # ... but k is called with ANOTHER tensor: the set is dissolved, the per-layer path answers
    assert torch.allclose(got_k, want_k, atol=1e-6) and q._sib is None and siblings.stats()["dissolved"] == 1
    assert yq.shape == (3, 7, 64)
    # an in-place write between the calls is the same thing (version counter)
    attn(h)
    assert attn._mods[0]._sib is not None
    attn.to_q(h)
    with torch.no_grad():
        h.add_(1.0)
    siblings.enable(False)
    want_k = attn.to_k(h)
    siblings.enable(True)
    assert torch.allclose(attn.to_k(h), want_k, atol=1e-6) and attn._mods[0]._sib is None


def test_restore_forgets_the_set_and_dropout_variants_stay_on_the_per_layer_path(host_eligible):
    torch.manual_seed(2)
    attn = Attn()
    attn._mods = adapt(attn)
    h = torch.randn(2, 4, 64)
    attn(h)
    q, k, v, _ = attn._mods
    assert q._sib is k._sib is v._sib is not None
    k.restore()
    assert q._sib is None and k._sib is None and v._sib is None
    k.apply_to()
    attn(h)
    assert q._sib is not None
    v.rank_dropout, v.training = 0.5, True       # v leaves the plain path: the leader must not compute it
    y = attn.to_q(h)
    assert q._sib is None and y.shape == (2, 4, 64)


def test_not_more_than_four_members_and_only_equal_shapes(host_eligible):
    torch.manual_seed(3)
    layers = [nn.Linear(64, 64) for _ in range(6)] + [nn.Linear(64, 128)]
    mods = []
    for i, l in enumerate(layers):
        m = LokrModule(f"l{i}", l, 1.0, 10000, 1, factor=4)
        m.apply_to()
        mods.append(m)
    x = torch.randn(5, 64)
    for l in layers:
        l(x)
    assert len(mods[0]._sib.members) == 4 and mods[4]._sib is not mods[0]._sib
    assert mods[6]._sib is None or all(r() is not mods[6] for r in mods[0]._sib.members)


def test_locon_and_low_rank_lokr_modules_form_sets_of_their_own_kind(host_eligible):
    """the mechanism is algorithm-agnostic (SiblingMixin): LoCon sets run ops.locon_linear_group, low-rank LoKr sets
    ops.lokr_linear_lr_group; modules of different algorithms or factor shapes never share a set"""
    torch.manual_seed(4)
    g = torch.Generator().manual_seed(4)
    for make in (lambda n, l: LoConModule(n, l, 1.0, 4, 2), lambda n, l: LokrModule(n, l, 1.0, 2, 1, factor=4)):
        attn = Attn()
        mods = []
        for name, layer in attn.named_modules():
            if isinstance(layer, nn.Linear):
                m = make(name, layer)
                with torch.no_grad():
                    for p in m.parameters():
                        p.copy_(torch.randn(p.shape, generator=g) * 0.2)
                m.apply_to()
                mods.append(m)
        h = torch.randn(2, 6, 64)
        siblings.enable(False)
        want = attn(h)
        siblings.enable(True)
        attn(h)
        assert mods[0]._sib is not None and len(mods[0]._sib.members) == 3, type(mods[0]).__name__
        before = siblings.stats()["launches"]
        got = attn(h)
        assert siblings.stats()["launches"] == before + 1 and torch.allclose(got, want, atol=1e-6)
        if isinstance(mods[0], LokrModule):
            assert not mods[0].use_w2  # rank 2 < 16 / 2: the low-rank pair
    # a LoCon layer and a LoKr layer called with the same tensor: two different keys, no set
    a, b = nn.Linear(64, 64), nn.Linear(64, 64)
    ma, mb = LoConModule("a", a, 1.0, 4, 2), LokrModule("b", b, 1.0, 10000, 1, factor=4)
    ma.apply_to(); mb.apply_to()
    x = torch.randn(3, 64)
    a(x); b(x)
    assert ma._sib is None and mb._sib is None


def test_fp32_activations_under_autocast_are_on_by_default():
    """siblings.enable(autocast=...): the eligibility test of the GPU path (activation_ok) -- host tensors never qualify, the flag is
    ON by default (round 6: sd-scripts' mixed precision hands fp32 LayerNorm outputs to to_q / to_k / to_v) and survives
    enable(True) / enable(False) unless named"""
    x = torch.randn(4, 64)
    assert siblings._STATE["autocast"] is True and not siblings.activation_ok(x) and not siblings.activation_ok(x.bfloat16())
    try:
        siblings.enable(True, autocast=False)
        assert siblings._STATE["autocast"] is False
        siblings.enable(False)
        siblings.enable(True)
        assert siblings._STATE["autocast"] is False and not siblings.activation_ok(x)   # (a host tensor all the same)
        # the set key tells autocast dtypes apart only for fp32 HIP tensors
        m = LokrModule("k", nn.Linear(64, 64), 1.0, 10000, 1, factor=4)
        assert siblings._key(m, x)[-1] is None
    finally:
        siblings.enable(True, autocast=True)
    assert siblings._STATE["autocast"] is True


def test_deepcopy_and_pickle_of_a_model_whose_sets_have_formed(host_eligible):
    """set membership (weakrefs + parked tensors) is not module state -- a deep copy / a pickled module drops it and
    learns its own sets on its first forward pass; the reference's modules support both"""
    import copy
    import io
    torch.manual_seed(0)
    block = Block()
    block._mods = adapt(block)
    h, ctx = torch.randn(2, 9, 64), torch.randn(2, 5, 32)
    run(block, h, ctx)
    want = run(block, h, ctx)
    q = block._mods[0]
    assert q._sib is not None and len(q._sib.members) == 3
    twin = copy.deepcopy(block)
    assert all("_sib" not in m.__dict__ and "_sib_key_cached" not in m.__dict__ for m in twin._mods)
    for _ in range(3):                       # learning pass, then two grouped passes of the COPY's own sets
        got = run(twin, h, ctx)
    assert twin._mods[0]._sib is not None and twin._mods[0]._sib is not q._sib
    assert torch.equal(got[0], want[0])
    assert q._sib is not None and len(q._sib.alive()) == 3          # the original's sets are untouched
    buf = io.BytesIO()
    torch.save(q, buf)                                                # whole-module pickling (torch.save(module), spawn)
    buf.seek(0)
    q2 = torch.load(buf, weights_only=False)
    assert q2._sib is None and torch.equal(q2.lokr_w2, q.lokr_w2)
    # a set emptied behind a member's back is dead, not an IndexError
    st = q._sib
    st.members = []
    assert st.alive() is None
    got = run(block, h, ctx)
    assert torch.equal(got[0], want[0])


def test_a_change_of_the_activation_shape_keeps_the_sets(host_eligible):
    """aspect-ratio buckets / a last partial batch change x.shape[:-1] only -- the kernels take any row count, so the
    sets stay; and a set that does dissolve re-forms completely (leader included) within the next two passes"""
    torch.manual_seed(0)
    block = Block()
    block._mods = adapt(block)
    ctx = torch.randn(2, 5, 32)
    run(block, torch.randn(2, 9, 64), ctx)
    for n in (9, 7, 12, 9):
        run(block, torch.randn(2, n, 64), ctx)
    st = siblings.stats()
    assert st["dissolved"] == 0 and st["sets"] == 2 and st["launches"] == 8, st
    q = block._mods[0]
    q._sib.dissolve()
    run(block, torch.randn(2, 9, 64), ctx)   # learning again (the leader records itself right after the dissolve)
    assert q._sib is not None and [r().lora_name for r in q._sib.members] == ["attn1_to_q", "attn1_to_k", "attn1_to_v"]
