"""LoHa functional API (mirror of lycoris/functional/loha.py).

weights tuple = (w1d, w1u, w2d, w2u, t1, t2) with w*d:[r, I*k*k] ("b" factors) and w*u:[O, r] ("a" factors);
gamma is the final multiplier (a float or a 0-d tensor, the reference modules pass a tensor).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .general import conv3d_aten, conv_args


def weight_gen(org_weight: torch.Tensor, rank: int, tucker: bool = True):
    """(w1d, w1u, w2d, w2u, t1, t2) initialised as functional/loha.py:86-116."""
    out_dim, in_dim, *k = org_weight.shape
    if k and tucker and any(i != 1 for i in k):  # functional/loha.py:94-108: cores [r, r, *k], "u" factors [r, O], "d" [r, I]
        w1d, w1u = torch.empty(rank, in_dim), torch.zeros(rank, out_dim)
        w2d, w2u = torch.empty(rank, in_dim), torch.empty(rank, out_dim)
        t1, t2 = torch.empty(rank, rank, *k), torch.empty(rank, rank, *k)
        for t in (t1, t2):
            nn.init.normal_(t, std=0.1)
        nn.init.normal_(w1d, std=1)
        nn.init.normal_(w2d, std=1)
        nn.init.normal_(w2u, std=0.1)
        return w1d, w1u, w2d, w2u, t1, t2
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
    if t1 is not None:  # HadaWeightTucker (functional/loha.py:33-75): rebuild_k = w_k_u^T @ fold(t_k, w_k_d)
        rank, out_dim = w1u.shape
        fold = lambda t, wd: torch.einsum("ij...,jq->iq...", t, wd) if (not t.is_cuda or t.dim() != 4) else ops.tucker_core(t, wd)
        dw = (w1u.t() @ fold(t1, w1d).flatten(1)) * (w2u.t() @ fold(t2, w2d).flatten(1))
        return (dw * gamma).reshape(out_dim, w1d.shape[1], *t1.shape[2:])
    rank = w1d.shape[0]
    out_dim = w1u.shape[0]
    dw = (w1u.reshape(out_dim, rank) @ w1d.reshape(rank, -1)) * (w2u.reshape(out_dim, rank) @ w2d.reshape(rank, -1))
    return (dw * gamma).reshape(out_dim, *w1d.shape[1:])


def bypass_forward_diff(x, org_out, *weights, gamma=1.0, extra_args={}):
    """delta = op(x, dW) with dW rebuilt tile-by-tile on chip (functional/loha.py:150-165; works, unlike upstream D2)."""
    w1d, w1u, w2d, w2u, t1, t2 = weights
    g = _gamma_value(gamma)
    shape5 = extra_args.get("_conv_shape") if len(extra_args.get("_conv_shape") or ()) == 5 else None
    if (t1 is not None and t1.dim() == 5) or w1d.dim() == 5 or shape5:  # nn.Conv3d weights: F.conv3d(x, dW) in ATen ops
        dw = diff_weight(w1d, w1u, w2d, w2u, t1, t2, gamma=gamma)  # (gamma may be a tensor: autograd-visible here)
        return conv3d_aten(x, dw.reshape(shape5) if shape5 else dw, extra_args)
    if t1 is not None:  # Tucker: the plain form on (w_u^T, fold(t, w_d)), cores folded by csrc/tucker.h
        shape = (w1u.shape[1], w1d.shape[1], *t1.shape[2:])
        stride, padding, dilation = conv_args({k: v for k, v in extra_args.items() if not k.startswith("_")})
        return ops.loha_conv2d(x, w1u.t(), ops.tucker_core(t1, w1d).flatten(1), w2u.t(), ops.tucker_core(t2, w2d).flatten(1), g,
                               shape, stride, padding, dilation)
    if w1d.dim() == 2 and not extra_args.get("_conv_shape"):
        return ops.loha_linear(x, w1u, w1d, w2u, w2d, g)
    shape = extra_args.get("_conv_shape") or (w1u.shape[0], *w1d.shape[1:])
    stride, padding, dilation = conv_args({k: v for k, v in extra_args.items() if not k.startswith("_")})
    return ops.loha_conv2d(x, w1u, w1d, w2u, w2d, g, tuple(shape), stride, padding, dilation)
