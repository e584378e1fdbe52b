"""Host-side helpers shared by the adapter algorithms (mirror of lycoris/functional/general.py)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

# weight.dim() -> the dense op the reference evaluates dW with (functional/general.py:6); Linear and Conv2d (Conv1d as its twin) run on
# the kernels here, this table is what the Conv3d / weight-space paths and callers of the functional API index
FUNC_LIST = [None, None, F.linear, F.conv1d, F.conv2d, F.conv3d]


def factorization(dimension: int, factor: int = -1) -> tuple[int, int]:
    """(m, n) with m * n == dimension and m <= n, the split LoKr uses for its Kronecker factors.

    Same contract as the reference's ``factorization`` (functional/general.py:14-56), pinned by
    tests/golden/factorization.json: an exact divisor ``factor`` is used as is; otherwise the most balanced
    divisor pair whose small side does not exceed ``factor`` (no limit when ``factor`` is negative).
    """
    dimension, factor = int(dimension), int(factor)
    if factor > 0 and dimension % factor == 0:
        pair = (factor, dimension // factor)
        return (min(pair), max(pair))
    limit = dimension if factor < 0 else factor
    small, large = 1, dimension
    budget = small + large  # the reference compares every candidate with 1 + dimension, never with the best so far
    while small < large:
        nxt = small + 1
        while dimension % nxt:
            nxt += 1
        if nxt + dimension // nxt > budget or nxt > limit:
            break
        small, large = nxt, dimension // nxt
    return (min(small, large), max(small, large))


def conv3d_aten(x: torch.Tensor, dw: torch.Tensor, extra_args: dict | None) -> torch.Tensor:
    """F.conv3d(x, dW): the reference's own evaluation for 5-D weights (FUNC_LIST[weight.dim()], functional/general.py:6), ATen ops on
    any device -- SURVEY 8a row a2 keeps Conv3d off the kernels.  Promoted dtype, one rounding to x's (composite.py's rule)."""
    ea = {k: v for k, v in (extra_args or {}).items() if k in ("stride", "padding", "dilation", "groups")}
    ct = torch.promote_types(x.dtype, dw.dtype)
    return torch.nn.functional.conv3d(x.to(ct), dw.to(ct), None, **ea).to(x.dtype)


def rebuild_tucker(t: torch.Tensor, wa: torch.Tensor, wb: torch.Tensor) -> torch.Tensor:
    """W[p, q, ...] = sum_ij t[i, j, ...] wa[i, p] wb[j, q]  (functional/general.py:9-11) = wa^T @ fold(t, wb): on the device the
    fold is the tucker_core kernel (csrc/tucker.h, differentiable); CPU tensors (offline tools) take the einsum."""
    if t.is_cuda and t.dim() == 4:
        from .. import ops
        fold = ops.tucker_core(t, wb)
        return (wa.t() @ fold.flatten(1)).reshape(wa.shape[1], wb.shape[1], *t.shape[2:])
    return torch.einsum("ij...,ip,jq->pq...", t, wa, wb)


def conv_args(extra_args: dict | None):
    """Normalise the conv ``kw_dict`` of the reference modules (modules/base.py:101-121) to int pairs."""
    ea = dict(extra_args or {})

    def pair(v, default):
        v = ea.get(v, default)
        if isinstance(v, (tuple, list)):
            if len(v) == 1:
                return int(v[0]), int(v[0])
            return int(v[0]), int(v[1])
        return int(v), int(v)

    groups = int(ea.get("groups", 1))
    if groups != 1:
        raise NotImplementedError("lycoris_amd: grouped convolutions are not supported by the adapter kernels")
    pad = ea.get("padding", 0)
    if isinstance(pad, str):
        raise NotImplementedError("lycoris_amd: string padding modes are not supported")
    return pair("stride", 1), pair("padding", 0), pair("dilation", 1)


def power2factorization(dimension: int, factor: int = -1):
    """(m, n) with m * n == dimension, m even, m <= factor and n a power of two -- the largest such m (functional/general.py:59-83, used
    by BOFT upstream); (None, 0) when there is none.  Pinned by tests/golden/power2factorization.json."""
    dimension, factor = int(dimension), int(factor)
    if factor == -1:
        factor = dimension
    best = 0
    for m in range(2, min(factor, dimension) + 1, 2):
        n = dimension // m
        if dimension % m == 0 and n & (n - 1) == 0:
            best = m
    return (best, dimension // best) if best else (None, 0)


def tucker_weight_from_conv(up: torch.Tensor, down: torch.Tensor, mid: torch.Tensor) -> torch.Tensor:
    """dW[o, i, ...] = sum_mn mid[m, n, ...] up[o, m] down[n, i] of a conv-CP LoCon triple given as conv weights (up / down 1x1:
    functional/general.py:86-89)"""
    return torch.einsum("mn...,om,ni->oi...", mid, up.flatten(1), down.flatten(1))


def tucker_weight(wa: torch.Tensor, wb: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """rebuild_tucker with the arguments in (wa, wb, t) order (functional/general.py:92-94)"""
    return rebuild_tucker(t, wa, wb)


def apply_dora_scale(org_weight: torch.Tensor, rebuild: torch.Tensor, dora_scale: torch.Tensor, scale: float) -> torch.Tensor:
    """W + ((W + dW) / ||W + dW||_in * dora_scale - W) * scale with the norm over each INPUT channel (functional/general.py:97-110: the
    wd_on_out=False form, no epsilon).  Host-side helper in plain tensor math; the modules' own DoRA path is modules/base.py."""
    merged = (org_weight + rebuild).to(dora_scale.dtype)
    norm = merged.transpose(0, 1).flatten(1).norm(dim=1).reshape(1, merged.shape[1], *([1] * (merged.dim() - 2)))
    return org_weight + (merged / norm * dora_scale - org_weight) * scale
