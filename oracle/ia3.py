"""(IA)^3 oracle (numpy, float64).  TEST INFRASTRUCTURE ONLY.

Rebuild-path semantics (NOT the bypass path, which also scales the bias -- SURVEY D9):
  dW = W * (w * multiplier) broadcast over out rows (train_on_input=False)
                               or in  cols (train_on_input=True)       modules/ia3.py:91-102
  y  = base + op(x, merged - W) = base + op(x, dW)                     modules/ia3.py:129-144
"""
import numpy as np

from .general import dense_backward, dense_forward, round_to


def _bshape(W, on_input):
    shp = [1] * W.ndim
    shp[1 if on_input else 0] = -1
    return shp


def diff_weight(W, w, multiplier=1.0, on_input=False):
    W = np.asarray(W, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64).reshape(-1)
    return W * (w * multiplier).reshape(_bshape(W, on_input))


def forward(x, W, w, multiplier=1.0, on_input=False, conv_args=None):
    return dense_forward(x, diff_weight(W, w, multiplier, on_input), conv_args)


def backward(x, g, W, w, multiplier=1.0, on_input=False, conv_args=None):
    """Returns (dx, d_w) with d_w shaped like ``w``."""
    W = np.asarray(W, dtype=np.float64)
    dx, dW = dense_backward(x, diff_weight(W, w, multiplier, on_input), g, conv_args)
    axes = tuple(i for i in range(W.ndim) if i != (1 if on_input else 0))
    d_w = (dW * W).sum(axis=axes) * multiplier
    return dx, d_w.reshape(np.asarray(w).shape)


# ---- the reference's BYPASS formulation with its storage roundings (modules/ia3.py:114-125, diff=True) -------------------
# `_bypass_forward`:  x = x * weight (train_on_input);  out = org_forward(x);  out = out * weight (otherwise)  -- every
# intermediate is a tensor of the activation dtype, i.e. rounded when the network runs in 16 bits.  ``store`` names that
# dtype ("bf16" / "f16" / None = exact).  The layer bias is left out: the native path follows the rebuild semantics for it
# (SURVEY D9), the cases that use this function are bias-free (SDXL to_k / to_v) or on the input side.
def bypass_forward(x, W, w, multiplier=1.0, on_input=False, conv_args=None, store=None):
    """Returns (delta, mid): mid is the rounded intermediate (the scaled input, or the frozen layer's output)."""
    x = np.asarray(x, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    wm = (np.asarray(w, dtype=np.float64).reshape(-1) * multiplier)
    r = (lambda a: a) if store is None else (lambda a: round_to(a, store))
    if on_input:
        shp = [1] * x.ndim
        shp[1 if W.ndim == 4 else -1] = -1
        mid = r(x * wm.reshape(shp))                       # x * weight                          ia3.py:116-117
        return dense_forward(mid, W, conv_args), mid       # org_forward(x)                      ia3.py:118
    mid = r(dense_forward(x, W, conv_args))                # org_forward(x) (stored in `store`)  ia3.py:118
    shp = [1] * mid.ndim
    shp[1 if W.ndim == 4 else -1] = -1
    return mid * wm.reshape(shp), mid                      # out * weight                        ia3.py:119-120


def bypass_backward(x, g, W, w, multiplier=1.0, on_input=False, conv_args=None, store=None):
    """Adjoint of bypass_forward with the same storage roundings: returns (dx, d_w)."""
    x = np.asarray(x, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    wm = w.reshape(-1) * multiplier
    r = (lambda a: a) if store is None else (lambda a: round_to(a, store))
    cd = 1 if W.ndim == 4 else -1
    if on_input:
        shp = [1] * x.ndim
        shp[cd] = -1
        mid = r(x * wm.reshape(shp))
        dmid, _ = dense_backward(mid, W, g, conv_args)     # gradient of the frozen op w.r.t. its (scaled) input
        dmid = r(dmid)                                     # ... a tensor of the activation dtype
        axes = tuple(i for i in range(x.ndim) if i != (cd % x.ndim))
        d_w = (dmid * x).sum(axis=axes) * multiplier
        return dmid * wm.reshape(shp), d_w.reshape(w.shape)
    mid = r(dense_forward(x, W, conv_args))
    shp = [1] * mid.ndim
    shp[cd] = -1
    axes = tuple(i for i in range(mid.ndim) if i != (cd % mid.ndim))
    d_w = (g * mid).sum(axis=axes) * multiplier
    dmid = r(g * wm.reshape(shp))                          # gradient handed to the frozen layer's backward
    dx, _ = dense_backward(x, W, dmid, conv_args)
    return dx, d_w.reshape(w.shape)
