"""GPU: dropout variants of the native modules -- applied around the kernels, never by rebuilding dW.

Semantics follow the reference's rebuild path: `module_dropout` skips the adapter for the call (locon.py:310-312,
lokr.py:544-546), `rank_dropout` drops rows of dW = output channels of the delta, optionally rescaled by the keep
rate (locon.py:210-217, loha.py:220-225, lokr.py:375-380), plain `dropout` acts on the delta in LoCon's bypass mode
(locon.py:304).  The random masks cannot be bit-identical to upstream's (different generators); they are reproduced here
by re-seeding torch and drawing the same calls."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_dw(algo, mod, shape):
    """float64 dW of the module's current factors, straight from the oracle (independent of the native kernels)."""
    f = {n: p.detach().double().cpu().numpy() for n, p in mod.named_parameters()}
    if algo == "locon":
        return oracle.locon.diff_weight(f["lora_down.weight"], f["lora_up.weight"], mod.scale).reshape(shape)
    if algo == "loha":
        return oracle.loha.diff_weight(f["hada_w1_a"], f["hada_w1_b"], f["hada_w2_a"], f["hada_w2_b"], mod.scale, shape)
    return oracle.lokr.diff_weight(w1=f["lokr_w1"], w2=f["lokr_w2"], scale=mod.scale, kshape=tuple(shape[2:])).reshape(shape)


def _expected_delta(x, dw64, row_mask, conv):
    """The reference's rebuild path with rank_dropout (modules/locon.py:210-217, loha.py:220-225, lokr.py:375-380):
    rows of dW are multiplied by the mask, then delta = op(x, dW)."""
    m = np.asarray(row_mask, dtype=np.float64).reshape(-1, *([1] * (dw64.ndim - 1)))
    ca = {"stride": 1, "padding": 1, "dilation": 1} if conv else None
    return oracle.general.dense_forward(x.detach().double().cpu().numpy(), dw64 * m, ca)


def _close(got, want64, tol=2e-5):
    return oracle.general.rel_err(got.detach().double().cpu().numpy(), want64) < tol


def _mods():
    from lycoris_amd.modules import LoConModule, LohaModule, LokrModule
    return {"locon": (LoConModule, dict(lora_dim=4, alpha=2)), "loha": (LohaModule, dict(lora_dim=4, alpha=2)),
            "lokr": (LokrModule, dict(lora_dim=100000, alpha=1, factor=4))}


def _build(algo, layer, **kw):
    cls, args = _mods()[algo]
    mod = cls("m", layer, 1.0, args["lora_dim"], args["alpha"], **{k: v for k, v in args.items() if k == "factor"}, **kw)
    mod.to(DEV)
    with torch.no_grad():  # non-zero factors (the default init makes every delta 0)
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * 0.2)
    return mod


@pytest.mark.parametrize("algo", ["locon", "loha", "lokr"])
@pytest.mark.parametrize("conv", [False, True])
def test_rank_and_module_dropout(algo, conv):
    torch.manual_seed(3)
    layer = (nn.Conv2d(32, 64, 3, padding=1) if conv else nn.Linear(32, 64)).to(DEV)
    x = torch.randn(2, 32, 6, 6, device=DEV) if conv else torch.randn(5, 32, device=DEV)
    base = layer(x)
    plain = _build(algo, layer)
    state = {k: v.clone() for k, v in plain.state_dict().items()}
    plain.apply_to()
    delta = layer(x) - base  # no dropout
    plain.restore()
    assert float(delta.detach().abs().max()) > 0
    dw64 = _oracle_dw(algo, plain, tuple(layer.weight.shape))
    assert _close(delta, _expected_delta(x, dw64, np.ones(64), conv), 1e-4)  # delta/base are fp32 differences

    for scale in (False, True):
        mod = _build(algo, layer, rank_dropout=0.5, rank_dropout_scale=scale)
        mod.load_state_dict(state)
        mod.train()
        mod.apply_to()
        torch.manual_seed(11)
        out = layer(x)
        mod.restore()
        torch.manual_seed(11)
        mask = (torch.rand(64, device=DEV) > 0.5).float()
        assert 0 < int(mask.sum()) < 64
        if scale:
            mask = mask / mask.mean()
        # independent expectation: the ORACLE's dW with its rows masked (not the native delta re-masked)
        want = _expected_delta(x, dw64, mask.cpu().numpy(), conv)
        assert _close(out - base, want, 1e-4), oracle.general.rel_err((out - base).detach().double().cpu().numpy(), want)
        mod.eval()  # inference: no dropout
        mod.apply_to()
        assert torch.allclose(layer(x) - base, delta, rtol=1e-4, atol=1e-5)
        mod.restore()

    mod = _build(algo, layer, module_dropout=1.0)
    mod.load_state_dict(state)
    mod.train()
    mod.apply_to()
    assert torch.equal(layer(x), base)  # adapter skipped
    mod.eval()
    assert torch.allclose(layer(x) - base, delta, rtol=1e-4, atol=1e-5)
    mod.restore()


def test_locon_bypass_dropout_acts_on_the_delta():
    torch.manual_seed(4)
    layer = nn.Linear(32, 64).to(DEV)
    x = torch.randn(7, 32, device=DEV)
    base = layer(x)
    plain = _build("locon", layer)
    state = {k: v.clone() for k, v in plain.state_dict().items()}
    plain.apply_to()
    delta = layer(x) - base
    plain.restore()
    mod = _build("locon", layer, dropout=0.25, bypass_mode=True)
    mod.load_state_dict(state)
    mod.train()
    mod.apply_to()
    torch.manual_seed(5)
    out = layer(x)
    mod.restore()
    torch.manual_seed(5)
    keep = torch.nn.functional.dropout(torch.ones_like(delta), 0.25, training=True)  # same generator calls: the mask
    dw64 = _oracle_dw("locon", plain, tuple(layer.weight.shape))
    want = _expected_delta(x, dw64, np.ones(64), False) * keep.double().cpu().numpy()   # locon.py:304 on the oracle delta
    assert _close(out - base, want, 1e-4)
    # the rebuild path (bypass_mode unset) ignores plain dropout, as upstream does
    mod2 = _build("locon", layer, dropout=0.25)
    mod2.load_state_dict(state)
    mod2.train()
    mod2.apply_to()
    assert torch.allclose(layer(x) - base, delta, rtol=1e-4, atol=1e-5)
    mod2.restore()
