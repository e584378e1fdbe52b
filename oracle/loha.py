"""LoHa oracle (numpy, float64).  TEST INFRASTRUCTURE ONLY.

  dW = ((w1a @ w1b) * (w2a @ w2b)) * scale                 functional/loha.py:10-15 (HadaWeight.forward)
  backward of the Hadamard product                          functional/loha.py:18-30 (HadaWeight.backward)
  conv (non-Tucker): w*_b is [r, I*kh*kw], dW viewed [O,I,kh,kw]   modules/loha.py:66-99, functional/loha.py:136-147
  y = base + op(x, dW * scalar * multiplier)                modules/loha.py:301-322

``round_dw=<dtype>`` restates the cast of the module's rebuild path,
``diff_weight = self.get_weight(self.shape).to(base_weight.dtype)`` (modules/loha.py:310): dW (scale included,
HadaWeight.forward multiplies it in, functional/loha.py:13-15) is rounded ONCE to the frozen weight's dtype before the
dense op and before the dense op's adjoint (dx = g @ round(dW)).  The factor gradients are unaffected by the cast in
exact arithmetic (autograd's ``.to`` backward is the identity), so ``backward`` only changes ``dx``.
"""
import numpy as np

from .general import dense_backward, dense_forward, round_to


def diff_weight(w1a, w1b, w2a, w2b, scale=1.0, shape=None, round_dw=None):
    w1a, w1b, w2a, w2b = (np.asarray(t, dtype=np.float64) for t in (w1a, w1b, w2a, w2b))
    dw = (w1a @ w1b.reshape(w1b.shape[0], -1)) * (w2a @ w2b.reshape(w2b.shape[0], -1)) * scale
    if round_dw is not None:
        dw = round_to(dw, round_dw)  # modules/loha.py:310
    return dw if shape is None else dw.reshape(shape)


def forward(x, w1a, w1b, w2a, w2b, scale=1.0, shape=None, conv_args=None, round_dw=None):
    return dense_forward(x, diff_weight(w1a, w1b, w2a, w2b, scale, shape, round_dw), conv_args)


def factor_grads(dW, w1a, w1b, w2a, w2b, scale=1.0):
    """(d_w1a, d_w1b, d_w2a, d_w2b) from the dense gradient w.r.t. dW: HadaWeight.backward (functional/loha.py:18-30)"""
    w1a, w1b, w2a, w2b = (np.asarray(t, dtype=np.float64) for t in (w1a, w1b, w2a, w2b))
    G = np.asarray(dW, dtype=np.float64).reshape(w1a.shape[0], -1) * scale
    b1 = w1b.reshape(w1b.shape[0], -1)
    b2 = w2b.reshape(w2b.shape[0], -1)
    t1 = G * (w2a @ b2)
    t2 = G * (w1a @ b1)
    return (t1 @ b1.T, (w1a.T @ t1).reshape(w1b.shape), t2 @ b2.T, (w2a.T @ t2).reshape(w2b.shape))


def backward(x, g, w1a, w1b, w2a, w2b, scale=1.0, shape=None, conv_args=None, round_dw=None):
    """Returns (dx, d_w1a, d_w1b, d_w2a, d_w2b).  With ``round_dw`` dx is taken through the rounded weight (the dense
    op's adjoint sees what the dense op saw); the factor gradients come from G = g^T x and the UNROUNDED rebuilds, as
    HadaWeight.backward computes them from the fp32 factors (functional/loha.py:18-30)."""
    dx, dW = dense_backward(x, diff_weight(w1a, w1b, w2a, w2b, scale, shape, round_dw), g, conv_args)
    return (dx, *factor_grads(dW, w1a, w1b, w2a, w2b, scale))
