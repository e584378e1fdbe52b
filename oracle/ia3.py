"""(IA)^3 oracle (numpy, float64).  TEST INFRASTRUCTURE ONLY.

Rebuild-path semantics (NOT the bypass path, which also scales the bias -- SURVEY D9):
  dW = W * (w * multiplier) broadcast over out rows (train_on_input=False)
                               or in  cols (train_on_input=True)       modules/ia3.py:91-102
  y  = base + op(x, merged - W) = base + op(x, dW)                     modules/ia3.py:129-144
"""
import numpy as np

from .general import dense_backward, dense_forward


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
