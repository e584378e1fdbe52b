"""Host-side helpers shared by the adapter algorithms (mirror of lycoris/functional/general.py)."""
from __future__ import annotations

import torch


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
