"""LoHa functional API (mirror of lycoris/functional/loha.py).

weights tuple = (w1d, w1u, w2d, w2u, t1, t2) with w*d:[r, I*k*k] ("b" factors) and w*u:[O, r] ("a" factors);
gamma is the final multiplier (a float or a 0-d tensor, the reference modules pass a tensor).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .general import conv_args


def weight_gen(org_weight: torch.Tensor, rank: int, tucker: bool = True):
    """(w1d, w1u, w2d, w2u, t1, t2) initialised as functional/loha.py:86-116 (non-Tucker layout only)."""
    out_dim, in_dim, *k = org_weight.shape
    if k and tucker:
        raise NotImplementedError("lycoris_amd: Tucker LoHa is not on the native path yet; pass tucker=False")
    w1d = torch.empty(rank, in_dim)
    w1u = torch.zeros(out_dim, rank)
    w2d = torch.empty(rank, in_dim)
    w2u = torch.empty(out_dim, rank)
    nn.init.normal_(w1d, std=1)
    nn.init.normal_(w2d, std=1)
    nn.init.normal_(w2u, std=0.1)
    return w1d, w1u, w2d, w2u, None, None


def _gamma_value(gamma):
    return float(gamma.detach()) if isinstance(gamma, torch.Tensor) else float(gamma)


def diff_weight(*weights, gamma=1.0):
    """Materialise dW = ((w1u @ w1d) * (w2u @ w2d)) * gamma (functional/loha.py:119-147).  Off the hot path."""
    w1d, w1u, w2d, w2u, t1, t2 = weights
    if t1 is not None or t2 is not None:
        raise NotImplementedError("lycoris_amd: Tucker LoHa is not supported")
    rank = w1d.shape[0]
    out_dim = w1u.shape[0]
    dw = (w1u.reshape(out_dim, rank) @ w1d.reshape(rank, -1)) * (w2u.reshape(out_dim, rank) @ w2d.reshape(rank, -1))
    return (dw * gamma).reshape(out_dim, *w1d.shape[1:])


def bypass_forward_diff(x, org_out, *weights, gamma=1.0, extra_args={}):
    """delta = op(x, dW) with dW rebuilt tile-by-tile on chip (functional/loha.py:150-165; works, unlike upstream D2)."""
    w1d, w1u, w2d, w2u, t1, t2 = weights
    if t1 is not None or t2 is not None:
        raise NotImplementedError("lycoris_amd: Tucker LoHa is not supported")
    g = _gamma_value(gamma)
    if w1d.dim() == 2 and not extra_args.get("_conv_shape"):
        return ops.loha_linear(x, w1u, w1d, w2u, w2d, g)
    shape = extra_args.get("_conv_shape") or (w1u.shape[0], *w1d.shape[1:])
    stride, padding, dilation = conv_args({k: v for k, v in extra_args.items() if not k.startswith("_")})
    return ops.loha_conv2d(x, w1u, w1d, w2u, w2d, g, tuple(shape), stride, padding, dilation)
