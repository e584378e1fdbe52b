"""LoCon functional API (mirror of lycoris/functional/locon.py; same names, argument order and gamma meaning).

weights tuple = (down, up, mid); gamma is the final multiplier (callers pass alpha / rank).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops
from .general import conv3d_aten, conv_args, rebuild_tucker


def weight_gen(org_weight: torch.Tensor, rank: int, tucker: bool = True):
    """(down, up, mid) initialised like the reference (functional/locon.py:10-34): kaiming down/mid, zero up."""
    out_dim, in_dim, *k = org_weight.shape
    if k and tucker:
        ones = [1] * len(k)
        down = torch.empty(rank, in_dim, *ones)
        up = torch.zeros(out_dim, rank, *ones)
        mid = torch.empty(rank, rank, *k)
        nn.init.kaiming_uniform_(down, a=math.sqrt(5))
        nn.init.kaiming_uniform_(mid, a=math.sqrt(5))
        return down, up, mid
    down = torch.empty(rank, in_dim)
    nn.init.kaiming_uniform_(down, a=math.sqrt(5))
    return down, torch.zeros(out_dim, rank), None


def diff_weight(*weights, gamma=1.0):
    """Materialise dW = (up * gamma) @ down (functional/locon.py:37-61).  Off the hot path (merge / export)."""
    down, up, mid = weights
    out_dim, rank = up.shape[0], up.shape[1]
    if mid is None:
        dw = (up.reshape(out_dim, rank) * gamma) @ down.reshape(rank, -1)
        return dw.reshape(out_dim, *down.shape[1:])
    dw = rebuild_tucker(mid, (up * gamma).reshape(out_dim, rank).t(), down.reshape(rank, -1))
    return dw.reshape(out_dim, down.shape[1], *mid.shape[2:])


def bypass_forward_diff(x, org_out, *weights, gamma=1.0, extra_args={}):
    """delta = up(down(x)) * gamma on the HIP path (functional/locon.py:64-85).  ``org_out`` is unused, as upstream."""
    down, up, mid = weights
    if down.dim() == 5:  # nn.Conv3d weights: F.conv3d(x, dW) in ATen ops
        return conv3d_aten(x, diff_weight(down, up, mid, gamma=gamma), extra_args)
    if mid is not None:  # conv-CP form: fold the k x k core into the 1x1 down-projection (csrc/tucker.h), then the plain path
        down = ops.tucker_core(mid, down)
    if down.dim() == 2:
        return ops.locon_linear(x, down, up, gamma)
    if down.dim() == 4:
        stride, padding, dilation = conv_args(extra_args)
        return ops.locon_conv2d(x, down, up, gamma, stride, padding, dilation)
    raise NotImplementedError("lycoris_amd: LoCon native path covers Linear and Conv2d")
