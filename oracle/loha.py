"""LoHa oracle (numpy, float64).  TEST INFRASTRUCTURE ONLY.

  dW = ((w1a @ w1b) * (w2a @ w2b)) * scale                 functional/loha.py:10-15 (HadaWeight.forward)
  backward of the Hadamard product                          functional/loha.py:18-30 (HadaWeight.backward)
  conv (non-Tucker): w*_b is [r, I*kh*kw], dW viewed [O,I,kh,kw]   modules/loha.py:66-99, functional/loha.py:136-147
  y = base + op(x, dW * scalar * multiplier)                modules/loha.py:301-322
"""
import numpy as np

from .general import dense_backward, dense_forward


def diff_weight(w1a, w1b, w2a, w2b, scale=1.0, shape=None):
    w1a, w1b, w2a, w2b = (np.asarray(t, dtype=np.float64) for t in (w1a, w1b, w2a, w2b))
    dw = (w1a @ w1b.reshape(w1b.shape[0], -1)) * (w2a @ w2b.reshape(w2b.shape[0], -1)) * scale
    return dw if shape is None else dw.reshape(shape)


def forward(x, w1a, w1b, w2a, w2b, scale=1.0, shape=None, conv_args=None):
    return dense_forward(x, diff_weight(w1a, w1b, w2a, w2b, scale, shape), conv_args)


def factor_grads(dW, w1a, w1b, w2a, w2b, scale=1.0):
    """(d_w1a, d_w1b, d_w2a, d_w2b) from the dense gradient w.r.t. dW: HadaWeight.backward (functional/loha.py:18-30)"""
    w1a, w1b, w2a, w2b = (np.asarray(t, dtype=np.float64) for t in (w1a, w1b, w2a, w2b))
    G = np.asarray(dW, dtype=np.float64).reshape(w1a.shape[0], -1) * scale
    b1 = w1b.reshape(w1b.shape[0], -1)
    b2 = w2b.reshape(w2b.shape[0], -1)
    t1 = G * (w2a @ b2)
    t2 = G * (w1a @ b1)
    return (t1 @ b1.T, (w1a.T @ t1).reshape(w1b.shape), t2 @ b2.T, (w2a.T @ t2).reshape(w2b.shape))


def backward(x, g, w1a, w1b, w2a, w2b, scale=1.0, shape=None, conv_args=None):
    """Returns (dx, d_w1a, d_w1b, d_w2a, d_w2b)."""
    dx, dW = dense_backward(x, diff_weight(w1a, w1b, w2a, w2b, scale, shape), g, conv_args)
    return (dx, *factor_grads(dW, w1a, w1b, w2a, w2b, scale))
