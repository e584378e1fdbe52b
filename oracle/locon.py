"""LoCon oracle (numpy, float64).  TEST INFRASTRUCTURE ONLY.

Rebuild-path semantics of the reference:
  dW = (up * gamma).reshape(O, r) @ down.reshape(r, I*kh*kw)      functional/locon.py:37-61
  y  = base + op(x, dW * scalar * multiplier)                      modules/locon.py:198-219, 309-332
``scale`` below is the *final* multiplier (alpha/r * scalar * multiplier).
"""
import numpy as np

from .general import dense_backward, dense_forward


def diff_weight(down, up, scale=1.0):
    down = np.asarray(down, dtype=np.float64)
    up = np.asarray(up, dtype=np.float64)
    r = down.shape[0]
    O = up.shape[0]
    dw = (up.reshape(O, r) * scale) @ down.reshape(r, -1)
    return dw.reshape(O, *down.shape[1:])


def forward(x, down, up, scale=1.0, conv_args=None):
    return dense_forward(x, diff_weight(down, up, scale), conv_args)


def factor_grads(dW, down, up, scale=1.0):
    """(d_down, d_up) from the dense gradient w.r.t. dW = (up * scale) @ down  (autograd through locon.py:198-219)"""
    down = np.asarray(down, dtype=np.float64)
    up = np.asarray(up, dtype=np.float64)
    r, O = down.shape[0], up.shape[0]
    dW2 = np.asarray(dW, dtype=np.float64).reshape(O, -1) * scale
    d_up = (dW2 @ down.reshape(r, -1).T).reshape(up.shape)
    d_down = (up.reshape(O, r).T @ dW2).reshape(down.shape)
    return d_down, d_up


def backward(x, g, down, up, scale=1.0, conv_args=None):
    """Returns (dx, d_down, d_up) of sum(g * forward(x, ...))."""
    dx, dW = dense_backward(x, diff_weight(down, up, scale), g, conv_args)
    d_down, d_up = factor_grads(dW, down, up, scale)
    return dx, d_down, d_up
