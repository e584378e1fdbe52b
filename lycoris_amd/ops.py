"""Autograd-level wrappers over the C ABI (one Function per fused forward/backward pair).

Shapes follow the kernels: activations are flattened to [M, features] (Linear) and must be contiguous;
factors are fp32 (16-bit factors are up-converted -- they are tiny); factor gradients come back in the
factor's dtype.  ``alpha`` is a Python float (scale * multiplier); a learnable ``scalar`` gate is folded
into the first factor by the caller so its gradient flows through ordinary autograd.
"""
from __future__ import annotations

import torch

from . import _native as N


def _f32c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _flat2d(x: torch.Tensor, feat: int) -> torch.Tensor:
    if x.shape[-1] != feat:
        raise ValueError(f"expected last dim {feat}, got {tuple(x.shape)}")
    x2 = x.reshape(-1, feat)
    return x2 if x2.is_contiguous() else x2.contiguous()


class _LokrLinear(torch.autograd.Function):
    """y = x @ (kron(w1, w2) * alpha)^T without materialising the Kronecker product."""

    @staticmethod
    def forward(ctx, x, w1, w2, alpha):
        N.require_device(x, "input")
        a, b = w1.shape
        c, d = w2.shape
        x2 = _flat2d(x, b * d)
        w1f, w2f = _f32c(w1), _f32c(w2)
        y = torch.empty((x2.shape[0], a * c), dtype=x.dtype, device=x.device)
        N.call("lyc_lokr_linear_fwd", N.ptr(x2), N.ptr(w1f), N.ptr(w2f), N.ptr(y), x2.shape[0], a, b, c, d,
               float(alpha), N.dtype_code(x.dtype), N.stream_ptr(x.device))
        ctx.save_for_backward(x2, w1, w2)
        ctx.alpha = float(alpha)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], a * c)

    @staticmethod
    def backward(ctx, g):
        x2, w1, w2 = ctx.saved_tensors
        a, b = w1.shape
        c, d = w2.shape
        g2 = _flat2d(g, a * c)
        w1f, w2f = _f32c(w1), _f32c(w2)
        need_x, need_w1, need_w2 = ctx.needs_input_grad[:3]
        M = x2.shape[0]
        dx = torch.empty_like(x2) if (need_x or need_w1) else None
        dw1 = torch.zeros_like(w1f) if need_w1 else None
        dw2 = torch.zeros_like(w2f) if need_w2 else None
        N.call("lyc_lokr_linear_bwd", N.ptr(g2), N.ptr(x2), N.ptr(w1f), N.ptr(w2f), N.ptr(dx), N.ptr(dw1),
               N.ptr(dw2), M, a, b, c, d, ctx.alpha, N.dtype_code(x2.dtype), N.stream_ptr(x2.device))
        return (dx.view(ctx.xshape) if need_x else None,
                dw1.to(w1.dtype) if need_w1 else None,
                dw2.to(w2.dtype) if need_w2 else None, None)


class _LoconLinear(torch.autograd.Function):
    """y = alpha * (x @ down^T) @ up^T with the rank-r intermediate kept in fp32."""

    @staticmethod
    def forward(ctx, x, down, up, alpha):
        N.require_device(x, "input")
        r, I = down.shape
        O = up.shape[0]
        x2 = _flat2d(x, I)
        df, uf = _f32c(down), _f32c(up)
        M = x2.shape[0]
        t = torch.zeros((M, r), dtype=torch.float32, device=x.device)
        y = torch.empty((M, O), dtype=x.dtype, device=x.device)
        N.call("lyc_locon_linear_fwd", N.ptr(x2), N.ptr(df), N.ptr(uf), N.ptr(t), N.ptr(y), M, I, O, r,
               float(alpha), N.dtype_code(x.dtype), N.stream_ptr(x.device))
        ctx.save_for_backward(x2, down, up, t)
        ctx.alpha = float(alpha)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, g):
        x2, down, up, t = ctx.saved_tensors
        r, I = down.shape
        O = up.shape[0]
        g2 = _flat2d(g, O)
        df, uf = _f32c(down), _f32c(up)
        need_x, need_d, need_u = ctx.needs_input_grad[:3]
        M = x2.shape[0]
        dt = torch.zeros((M, r), dtype=torch.float32, device=g.device)
        dx = torch.empty_like(x2) if need_x else None
        dd = torch.zeros_like(df) if need_d else None
        du = torch.zeros_like(uf) if need_u else None
        N.call("lyc_locon_linear_bwd", N.ptr(g2), N.ptr(x2), N.ptr(df), N.ptr(uf), N.ptr(t), N.ptr(dt), N.ptr(dx),
               N.ptr(dd), N.ptr(du), M, I, O, r, ctx.alpha, N.dtype_code(x2.dtype), N.stream_ptr(x2.device))
        return (dx.view(ctx.xshape) if need_x else None,
                dd.to(down.dtype) if need_d else None,
                du.to(up.dtype) if need_u else None, None)


def _chan_dims(t: torch.Tensor, chan_dim: int):
    C = t.shape[chan_dim]
    outer = 1
    for s in t.shape[:chan_dim]:
        outer *= s
    inner = 1
    for s in t.shape[chan_dim + 1:]:
        inner *= s
    return outer, C, inner


class _ChanAffine(torch.autograd.Function):
    """out = a * (s0 + w[c]*mult) - bias[c]*w[c]*mult   over the channel dimension ``chan_dim``.

    s0 = 1, bias = layer bias : (IA)^3 out-side  y = base + (base - bias) * w*mult
    s0 = 0, bias = None       : (IA)^3 in-side   x * (w*mult)
    """

    @staticmethod
    def forward(ctx, a, w, bias, s0, mult, chan_dim):
        N.require_device(a, "input")
        a = a.contiguous()
        chan_dim = chan_dim % a.dim()
        outer, C, inner = _chan_dims(a, chan_dim)
        wf = _f32c(w).reshape(-1)
        if wf.numel() != C:
            raise ValueError(f"(IA)^3 weight has {wf.numel()} entries, channel dim has {C}")
        bf = None if bias is None else _f32c(bias).reshape(-1)
        out = torch.empty_like(a)
        N.call("lyc_chan_scale", N.ptr(a), N.ptr(wf), N.ptr(bf), N.ptr(out), outer, C, inner, float(s0),
               float(mult), N.dtype_code(a.dtype), N.stream_ptr(a.device))
        ctx.save_for_backward(a, w, bias)
        ctx.cfg = (float(s0), float(mult), chan_dim)
        return out

    @staticmethod
    def backward(ctx, g):
        a, w, bias = ctx.saved_tensors
        s0, mult, chan_dim = ctx.cfg
        g = g.contiguous()
        outer, C, inner = _chan_dims(a, chan_dim)
        wf = _f32c(w).reshape(-1)
        bf = None if bias is None else _f32c(bias).reshape(-1)
        code, st = N.dtype_code(a.dtype), N.stream_ptr(a.device)
        da = dw = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(g)
            N.call("lyc_chan_scale", N.ptr(g), N.ptr(wf), None, N.ptr(da), outer, C, inner, s0, mult, code, st)
        if ctx.needs_input_grad[1]:
            dwf = torch.zeros(C, dtype=torch.float32, device=a.device)
            N.call("lyc_chan_reduce", N.ptr(g), N.ptr(a), N.ptr(bf), N.ptr(dwf), outer, C, inner, mult, code, st)
            dw = dwf.reshape(w.shape).to(w.dtype)
        return da, dw, None, None, None, None


def lokr_linear(x, w1, w2, alpha=1.0):
    return _LokrLinear.apply(x, w1, w2, alpha)


def locon_linear(x, down, up, alpha=1.0):
    return _LoconLinear.apply(x, down, up, alpha)


def chan_affine(a, w, bias=None, s0=0.0, mult=1.0, chan_dim=-1):
    return _ChanAffine.apply(a, w, bias, s0, mult, chan_dim)
