"""LoKr oracle (numpy, float64).  TEST INFRASTRUCTURE ONLY.

  w1 = w1 | w1a @ w1b ; w2 = w2 | (w2a @ w2b.view(r,-1)).view(c, d, *k)     functional/lokr.py:124-151
  dW = kron(w1[..., None, None], w2) * scale                                  functional/lokr.py:11-20 (make_kron)
       => dW[p*c+q, u*d+v, i, j] = w1[p,u] * w2[q,v,i,j] * scale
  y  = base + op(x, dW * scalar * multiplier)                                 modules/lokr.py:358-381, 543-566
``scale`` is the final multiplier.  The Tucker (t2) form is not covered yet.
"""
import numpy as np

from .general import dense_backward, dense_forward


def _full_factors(w1, w1a, w1b, w2, w2a, w2b, kshape):
    f1 = np.asarray(w1, dtype=np.float64) if w1 is not None else np.asarray(w1a, np.float64) @ np.asarray(w1b, np.float64)
    if w2 is not None:
        f2 = np.asarray(w2, dtype=np.float64)
    else:
        w2a = np.asarray(w2a, np.float64)
        w2b = np.asarray(w2b, np.float64)
        f2 = (w2a @ w2b.reshape(w2b.shape[0], -1)).reshape(w2a.shape[0], -1, *kshape)
    return f1, f2


def diff_weight(w1=None, w1a=None, w1b=None, w2=None, w2a=None, w2b=None, scale=1.0, kshape=()):
    f1, f2 = _full_factors(w1, w1a, w1b, w2, w2a, w2b, kshape)
    a, b = f1.shape
    c, d = f2.shape[:2]
    dw = np.einsum("pu,qv...->pquv...", f1, f2) * scale
    return dw.reshape(a * c, b * d, *f2.shape[2:])


def forward(x, w1=None, w1a=None, w1b=None, w2=None, w2a=None, w2b=None, scale=1.0, kshape=(), conv_args=None):
    return dense_forward(x, diff_weight(w1, w1a, w1b, w2, w2a, w2b, scale, kshape), conv_args)


def factor_grads(dW, w1=None, w1a=None, w1b=None, w2=None, w2a=None, w2b=None, scale=1.0, kshape=()):
    """dict of gradients for every factor that was given, from the dense gradient w.r.t. dW (torch.kron's backward,
    functional/lokr.py:11-20, then the low-rank products)"""
    f1, f2 = _full_factors(w1, w1a, w1b, w2, w2a, w2b, kshape)
    a, b = f1.shape
    c, d = f2.shape[:2]
    dW5 = np.asarray(dW, dtype=np.float64).reshape(a, c, b, d, -1) * scale
    d_f1 = np.einsum("pquvk,qvk->pu", dW5, f2.reshape(c, d, -1))
    d_f2 = np.einsum("pquvk,pu->qvk", dW5, f1).reshape(f2.shape)
    out = {}
    if w1 is not None:
        out["w1"] = d_f1
    else:
        out["w1a"] = d_f1 @ np.asarray(w1b, np.float64).T
        out["w1b"] = np.asarray(w1a, np.float64).T @ d_f1
    if w2 is not None:
        out["w2"] = d_f2
    else:
        w2a_ = np.asarray(w2a, np.float64)
        w2b_ = np.asarray(w2b, np.float64)
        df2 = d_f2.reshape(c, -1)
        out["w2a"] = df2 @ w2b_.reshape(w2b_.shape[0], -1).T
        out["w2b"] = (w2a_.T @ df2).reshape(w2b_.shape)
    return out


def backward(x, g, w1=None, w1a=None, w1b=None, w2=None, w2a=None, w2b=None, scale=1.0, kshape=(), conv_args=None):
    """Returns dict with dx and a gradient for every factor that was given."""
    dx, dW = dense_backward(x, diff_weight(w1, w1a, w1b, w2, w2a, w2b, scale, kshape), g, conv_args)
    out = {"dx": dx}
    out.update(factor_grads(dW, w1, w1a, w1b, w2, w2a, w2b, scale, kshape))
    return out
