"""ATen composite forms of the adapter ops, for tensors that are NOT on the HIP device.

BASELINE configs[0] ("LoCon rank=4 on a 3-layer nn.Linear MLP via standalone wrapper, CPU (plumbing; mirrors
example/standalone_example.py)") trains on the CPU; SURVEY 7 step 1 asks that the package "must still import and run" without a
GPU.  A CPU tensor therefore takes the same entry points of `lycoris_amd.ops` and lands here: the FACTORED evaluation of each
algorithm written with plain differentiable tensor ops (torch's autograd provides the backward), in the promoted dtype of
activation and factors, rounded once to the activation dtype -- the reference's functional forms
(lycoris/functional/locon.py:64-99, loha.py:10-30 + modules/loha.py:301-322, lokr.py:154-247, modules/ia3.py:91-144).

This is the device dispatch of the op, not a fallback for the HIP path: a tensor on `cuda` ALWAYS takes the HIP kernels and fails
loudly when the extension is missing (`_native.NativeLibraryError`); nothing here is reachable with a device tensor, and nothing
here touches `oracle/` (that is numpy float64 test infrastructure).  Pinned against the reference-generated golden vectors by
tests/test_cpu_composite.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _ct(x, *factors):
    """compute dtype: the promotion of the activation and factor dtypes (fp32 factors on a bf16 activation -> fp32)"""
    dt = x.dtype
    for f in factors:
        if f is not None:
            dt = torch.promote_types(dt, f.dtype)
    return dt


def lokr_linear(x, w1, w2, alpha=1.0, base=None):
    """(w1 (x) w2) x without the Kronecker product: x [.., b*d] -> [.., b, d] @ w2^T -> [.., b, c]; contract b with w1 -> [.., a, c]"""
    a, b = w1.shape
    c, d = w2.shape
    if x.shape[-1] != b * d:
        raise RuntimeError(f"adapter expects {b * d} input features, got {tuple(x.shape)}")
    ct = _ct(x, w1, w2)
    hb = x.to(ct).reshape(*x.shape[:-1], b, d) @ w2.to(ct).t()
    y = (w1.to(ct) @ hb).reshape(*x.shape[:-1], a * c) * alpha
    y = y.to(x.dtype)
    return y if base is None else base + y


def locon_linear(x, down, up, alpha=1.0):
    ct = _ct(x, down, up)
    return ((x.to(ct) @ down.to(ct).t()) @ up.to(ct).t() * alpha).to(x.dtype)


def _hada(w1a, w1b, w2a, w2b):
    return (w1a @ w1b) * (w2a @ w2b)


def loha_linear(x, w1a, w1b, w2a, w2b, alpha=1.0):
    """dW rebuilt (an [O, I] Hadamard product has no factored form), rounded once to the activation dtype as the reference does
    (modules/loha.py:310 `diff_weight.to(base_weight.dtype)`)"""
    dw = _hada(w1a, w1b, w2a, w2b) * alpha
    return F.linear(x, dw.to(x.dtype))


def chan_affine(a, w, bias=None, s0=0.0, mult=1.0, chan_dim=-1):
    """a * (s0 + w[c] * mult) - bias[c] * w[c] * mult over the channel dimension (the (IA)^3 forms, csrc/ia3_kernels.h)"""
    chan_dim = chan_dim % a.dim()
    shape = [1] * a.dim()
    shape[chan_dim] = a.shape[chan_dim]
    ct = _ct(a, w, bias)
    wv = w.to(ct).reshape(shape) * mult
    out = a.to(ct) * (s0 + wv)
    if bias is not None:
        out = out - bias.to(ct).reshape(shape) * wv
    return out.to(a.dtype)


def locon_conv2d(x, down, up, alpha, stride, padding, dilation):
    ct = _ct(x, down, up)
    t = F.conv2d(x.to(ct), down.to(ct), None, tuple(stride), tuple(padding), tuple(dilation))
    return (F.conv2d(t, up.to(ct).reshape(up.shape[0], up.shape[1], 1, 1)) * alpha).to(x.dtype)


def lokr_conv2d(x, w1, w2, alpha, stride, padding, dilation):
    """functional/lokr.py:154-247, the Conv2d branch: the b input-channel groups as batch entries through conv(w2), then the
    b -> a contraction with w1; input channel u*d + v, output channel p*c + q (the kron indexing)"""
    a, b = w1.shape
    c, d = w2.shape[:2]
    B, C, H, W = x.shape
    if C != b * d:
        raise RuntimeError(f"adapter expects {b * d} input channels, got {tuple(x.shape)}")
    ct = _ct(x, w1, w2)
    hb = F.conv2d(x.to(ct).reshape(B * b, d, H, W), w2.to(ct), None, tuple(stride), tuple(padding), tuple(dilation))
    Ho, Wo = hb.shape[-2:]
    y = torch.einsum("pu,nuchw->npchw", w1.to(ct), hb.reshape(B, b, c, Ho, Wo)).reshape(B, a * c, Ho, Wo) * alpha
    return y.to(x.dtype)


def loha_conv2d(x, w1a, w1b, w2a, w2b, alpha, shape, stride, padding, dilation):
    dw = (_hada(w1a, w1b.reshape(w1b.shape[0], -1), w2a, w2b.reshape(w2b.shape[0], -1)) * alpha).reshape(tuple(shape))
    return F.conv2d(x, dw.to(x.dtype), None, tuple(stride), tuple(padding), tuple(dilation))
