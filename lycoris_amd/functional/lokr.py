"""LoKr functional API (mirror of lycoris/functional/lokr.py).

weights tuple = (w1, w1a, w1b, w2, w2a, w2b, t2).  NB the reference's gamma convention for LoKr: gamma is *alpha*,
and the function divides by the rank it infers from the low-rank factors (scale = gamma / rank; when both w1 and
w2 are full matrices rank := gamma, i.e. scale = 1) -- functional/lokr.py:135-141, 171-172.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops
from .general import conv3d_aten, conv_args, factorization


def weight_gen(org_weight, rank, tucker=True, factor=-1, decompose_both=False, full_matrix=False,
               unbalanced_factorization=False):
    """(w1, w1a, w1b, w2, w2a, w2b, t2), shapes and inits as functional/lokr.py:23-121 (non-Tucker forms)."""
    out_dim, in_dim, *k = org_weight.shape
    in_m, in_n = factorization(in_dim, factor)
    out_l, out_k = factorization(out_dim, factor)
    if unbalanced_factorization:
        out_l, out_k = out_k, out_l
    w1 = w1a = w1b = w2 = w2a = w2b = None
    split_w1 = decompose_both and rank < max(out_l, in_m) / 2 and not (k and full_matrix)
    if split_w1:
        w1a, w1b = torch.empty(out_l, rank), torch.empty(rank, in_m)
        nn.init.kaiming_uniform_(w1a, a=math.sqrt(5))
        nn.init.kaiming_uniform_(w1b, a=math.sqrt(5))
    else:
        w1 = torch.empty(out_l, in_m)
        nn.init.kaiming_uniform_(w1, a=math.sqrt(5))
    if k:
        full_w2 = rank >= max(out_k, in_n) / 2 or full_matrix
    else:
        full_w2 = not (rank < max(out_k, in_n) / 2)
    if full_w2:
        w2 = torch.zeros(out_k, in_n, *k)
    else:
        if k and tucker and any(i != 1 for i in k):  # functional/lokr.py:88-101: core [r, r, *k], [r, out_k], [r, in_n]
            t2 = torch.empty(rank, rank, *k)
            w2a, w2b = torch.empty(rank, out_k), torch.zeros(rank, in_n)
            nn.init.kaiming_uniform_(t2, a=math.sqrt(5))
            nn.init.kaiming_uniform_(w2a, a=math.sqrt(5))
            return w1, w1a, w1b, w2, w2a, w2b, t2
        w2a = torch.empty(out_k, rank)
        w2b = torch.zeros(rank, in_n, *k)
        nn.init.kaiming_uniform_(w2a, a=math.sqrt(5))
    return w1, w1a, w1b, w2, w2a, w2b, None


def _resolve(weights, gamma):
    w1, w1a, w1b, w2, w2a, w2b, t = weights
    if w1a is not None:
        rank = w1a.shape[1]
    elif t is not None:
        rank = t.shape[0]
    elif w2a is not None:
        rank = w2a.shape[1]
    else:
        rank = gamma
    scale = gamma / rank
    f1 = w1 if w1 is not None else w1a @ w1b
    if w2 is not None:
        f2 = w2
    elif t is not None:  # rebuild_tucker(t, w2a, w2b) = w2a^T @ fold(t, w2b)   (functional/general.py:9-11, csrc/tucker.h)
        fold = ops.tucker_core(t, w2b) if (t.is_cuda and t.dim() == 4) else torch.einsum("ij...,jq->iq...", t, w2b)
        f2 = (w2a.t() @ fold.flatten(1)).reshape(w2a.shape[1], w2b.shape[1], *t.shape[2:])
    else:
        f2 = (w2a @ w2b.reshape(w2b.shape[0], -1)).reshape(w2a.shape[0], *w2b.shape[1:])
    return f1, f2, scale


def make_kron(w1, w2, scale):
    """kron(w1, w2) * scale with w1 broadcast over the kernel dims (functional/lokr.py:11-20).  Offline helper."""
    f1 = w1.reshape(*w1.shape, *([1] * (w2.dim() - w1.dim())))
    out = torch.kron(f1, w2.contiguous())
    return out if scale == 1 else out * scale


def diff_weight(*weights, gamma=1.0):
    """Materialise dW (functional/lokr.py:124-151).  Off the hot path (merge / export)."""
    f1, f2, scale = _resolve(weights, gamma)
    return make_kron(f1, f2, scale)


def bypass_forward_diff(h, org_out, *weights, gamma=1.0, extra_args={}):
    """delta = (w1 (x) w2) h * scale, Kronecker-factored on the HIP path (functional/lokr.py:154-247)."""
    f1, f2, scale = _resolve(weights, gamma)
    scale = float(scale)
    if f2.dim() == 2:
        return ops.lokr_linear(h, f1, f2, scale)
    if f2.dim() == 4:
        stride, padding, dilation = conv_args(extra_args)
        return ops.lokr_conv2d(h, f1, f2, scale, stride, padding, dilation)
    if f2.dim() == 5:  # nn.Conv3d weights: F.conv3d(x, kron(w1, w2)) in ATen ops
        return conv3d_aten(h, make_kron(f1, f2, scale), extra_args)
    raise NotImplementedError("lycoris_amd: LoKr covers Linear, Conv2d (kernels) and Conv3d (ATen)")
