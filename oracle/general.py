"""Shared oracle helpers (numpy, float64).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/lycoris/functional/general.py:
  * factorization      -> general.py:14-56
  * FUNC_LIST dispatch -> general.py:6 (F.linear / F.conv2d); conv arguments are
    the ``kw_dict`` of modules/base.py:101-121 (stride, padding, dilation, groups=1)
"""
from __future__ import annotations

import numpy as np


def factorization(dimension: int, factor: int = -1):
    """Split ``dimension`` into (m, n), m <= n, m*n == dimension.

    Restates general.py:14-56: if ``factor`` divides the dimension, the split is
    (factor, dimension/factor) ordered ascending; otherwise walk the divisors
    upward from 1 while the pair keeps getting more balanced (m + n does not
    grow) and m stays <= factor (factor < 0 means "no cap").
    """
    dimension = int(dimension)
    factor = int(factor)
    if factor > 0 and dimension % factor == 0:
        lo, hi = factor, dimension // factor
        return (lo, hi) if lo <= hi else (hi, lo)
    cap = dimension if factor < 0 else factor
    m, n = 1, dimension
    best_sum = m + n
    while m < n:
        cand = m + 1
        while dimension % cand:
            cand += 1
        other = dimension // cand
        if cand + other > best_sum or cand > cap:
            break
        m, n = cand, other
        # NB: the reference never updates its running "length" (general.py:43,49),
        # so the comparison is always against 1 + dimension.
    return (m, n) if m <= n else (n, m)


# ----------------------------------------------------------------------------
# dense ops: y = op(x, W) and their adjoints
# ----------------------------------------------------------------------------
def _pair(v):
    if isinstance(v, (tuple, list)):
        return int(v[0]), int(v[1])
    return int(v), int(v)


def _im2col(x, kh, kw, stride, padding, dilation):
    """x:[B,C,H,W] -> cols:[B, C*kh*kw, Ho*Wo] (channel-major, then kh, kw)."""
    B, C, H, W = x.shape
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    xp = np.zeros((B, C, H + 2 * ph, W + 2 * pw), dtype=x.dtype)
    xp[:, :, ph:ph + H, pw:pw + W] = x
    cols = np.empty((B, C, kh, kw, Ho, Wo), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            hs, ws = i * dh, j * dw
            cols[:, :, i, j] = xp[:, :, hs:hs + sh * (Ho - 1) + 1:sh, ws:ws + sw * (Wo - 1) + 1:sw]
    return cols.reshape(B, C * kh * kw, Ho * Wo), (Ho, Wo)


def _col2im(cols, xshape, kh, kw, stride, padding, dilation, Ho, Wo):
    B, C, H, W = xshape
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    xp = np.zeros((B, C, H + 2 * ph, W + 2 * pw), dtype=cols.dtype)
    cols = cols.reshape(B, C, kh, kw, Ho, Wo)
    for i in range(kh):
        for j in range(kw):
            hs, ws = i * dh, j * dw
            xp[:, :, hs:hs + sh * (Ho - 1) + 1:sh, ws:ws + sw * (Wo - 1) + 1:sw] += cols[:, :, i, j]
    return xp[:, :, ph:ph + H, pw:pw + W]


def dense_forward(x, w, conv_args=None):
    """op(x, W) of FUNC_LIST (general.py:6): F.linear for 2-D W, F.conv2d for 4-D W."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    if w.ndim == 2:
        return x @ w.T
    assert w.ndim == 4, "oracle covers Linear and Conv2d only"
    ca = conv_args or {}
    assert int(ca.get("groups", 1)) == 1
    O, C, kh, kw = w.shape
    cols, (Ho, Wo) = _im2col(x, kh, kw, ca.get("stride", 1), ca.get("padding", 0), ca.get("dilation", 1))
    y = np.matmul(w.reshape(O, -1), cols)  # [O, K] @ [B, K, P] -> [B, O, P] (BLAS: full-size layers take seconds)
    return y.reshape(x.shape[0], O, Ho, Wo)


def dense_backward(x, w, g, conv_args=None):
    """Adjoint of dense_forward: returns (dx, dW) for upstream gradient g."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    if w.ndim == 2:
        dx = g @ w
        dw = g.reshape(-1, g.shape[-1]).T @ x.reshape(-1, x.shape[-1])
        return dx, dw
    ca = conv_args or {}
    O, C, kh, kw = w.shape
    st, pd, dl = ca.get("stride", 1), ca.get("padding", 0), ca.get("dilation", 1)
    cols, (Ho, Wo) = _im2col(x, kh, kw, st, pd, dl)
    g2 = g.reshape(g.shape[0], O, Ho * Wo)
    dw = np.matmul(g2, cols.transpose(0, 2, 1)).sum(axis=0).reshape(w.shape)
    dcols = np.matmul(w.reshape(O, -1).T, g2)
    dx = _col2im(dcols, x.shape, kh, kw, st, pd, dl, Ho, Wo)
    return dx, dw


def tucker_core(t, wb):
    """Fold the Tucker core into the input-side factor: B[i, q, ...] = sum_j t[i, j, ...] wb[j, q].
    rebuild_tucker(t, wa, wb) (general.py:9-11: einsum "i j ..., i p, j r -> p r ...") == wa^T @ B."""
    t = np.asarray(t, dtype=np.float64)
    wb = np.asarray(wb, dtype=np.float64).reshape(t.shape[1], -1)
    return np.einsum("ij...,jq->iq...", t, wb)


def tucker_core_grads(dB, t, wb):
    """(d_t, d_wb) of sum(dB * tucker_core(t, wb))"""
    t = np.asarray(t, dtype=np.float64)
    wb2 = np.asarray(wb, dtype=np.float64).reshape(t.shape[1], -1)
    t3 = t.reshape(t.shape[0], t.shape[1], -1)
    dB3 = np.asarray(dB, dtype=np.float64).reshape(t.shape[0], wb2.shape[1], -1)
    d_t = np.einsum("iqk,jq->ijk", dB3, wb2).reshape(t.shape)
    d_wb = np.einsum("ijk,iqk->jq", t3, dB3)
    return d_t, d_wb.reshape(np.asarray(wb).shape)


def rebuild_tucker(t, wa, wb):
    """W[p, q, ...] = sum_ij t[i, j, ...] wa[i, p] wb[j, q]   (functional/general.py:9-11)"""
    return np.einsum("ij...,ip,jq->pq...", np.asarray(t, np.float64), np.asarray(wa, np.float64), np.asarray(wb, np.float64))


def round_to(a, dtype):
    """Round a float64 array the way ``tensor.to(dtype)`` rounds an fp32 tensor (round-to-nearest-even), back to float64.
    ``dtype``: "bf16" / "f16" / "f32" (or a name containing "bfloat16" / "float16" / "float32", e.g. str(torch.bfloat16)).
    The value passes through float32 first: the reference forms its weights in fp32 and THEN casts
    (``get_weight(...).to(base_weight.dtype)``, modules/loha.py:310)."""
    name = str(dtype)
    a32 = np.asarray(a, dtype=np.float64).astype(np.float32)
    if "bf16" in name or "bfloat16" in name:
        u = np.ascontiguousarray(a32).view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16          # RNE on the dropped 16 bits (finite values)
        return u.astype(np.uint32).view(np.float32).reshape(a32.shape).astype(np.float64)
    if "f16" in name or "float16" in name:
        with np.errstate(over="ignore"):
            return a32.astype(np.float16).astype(np.float64)
    if "f32" in name or "float32" in name:
        return a32.astype(np.float64)
    raise ValueError(f"round_to: unknown dtype {dtype!r}")


def rel_err(a, b):
    """Norm-wise relative error ||a-b|| / ||b|| (the metric of SURVEY 8d)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))
