"""DoRA (weight_decompose) oracle (numpy, float64).  TEST INFRASTRUCTURE ONLY.

apply_weight_decompose (modules/locon.py:239-260, loha.py:244-265, lokr.py:399-420) inside the rebuild forward
(locon.py:320-332):
    V      = W + dW                      dW = the algorithm's diff weight with scale * scalar (NO multiplier)
    n[ch]  = ||V||_ch + eps              ch = output row (wd_on_out) or input channel (norm over out, kh, kw)
    s[ch]  = multiplier * (dora_scale[ch] / n[ch] - 1) + 1
    y      = base + op(x, V * s - W)
"""
import numpy as np

from .general import dense_backward, dense_forward

EPS64 = float(np.finfo(np.float64).eps)


def _axes(V, on_out):
    ch = 0 if on_out else 1
    return ch, tuple(i for i in range(V.ndim) if i != ch)


def _bshape(V, ch):
    shp = [1] * V.ndim
    shp[ch] = -1
    return shp


def scale_vector(W, dW, dora_scale, multiplier=1.0, on_out=True, eps=EPS64):
    V = np.asarray(W, np.float64) + np.asarray(dW, np.float64)
    ch, axes = _axes(V, on_out)
    n = np.sqrt((V * V).sum(axis=axes)) + eps
    s = multiplier * (np.asarray(dora_scale, np.float64).reshape(-1) / n - 1.0) + 1.0
    return V, n, s


def delta_weight(W, dW, dora_scale, multiplier=1.0, on_out=True, eps=EPS64):
    V, n, s = scale_vector(W, dW, dora_scale, multiplier, on_out, eps)
    ch, _ = _axes(V, on_out)
    return V * s.reshape(_bshape(V, ch)) - np.asarray(W, np.float64)


def forward(x, W, dW, dora_scale, multiplier=1.0, on_out=True, conv_args=None, eps=EPS64):
    return dense_forward(x, delta_weight(W, dW, dora_scale, multiplier, on_out, eps), conv_args)


def backward(x, g, W, dW, dora_scale, multiplier=1.0, on_out=True, conv_args=None, eps=EPS64):
    """Returns (dx, d_dW, d_dora_scale): d_dW is the dense gradient w.r.t. dW (push it through the algorithm's
    factor_grads), d_dora_scale is shaped like dora_scale."""
    V, n, s = scale_vector(W, dW, dora_scale, multiplier, on_out, eps)
    ch, axes = _axes(V, on_out)
    bs = _bshape(V, ch)
    dx, G = dense_backward(x, V * s.reshape(bs) - np.asarray(W, np.float64), g, conv_args)   # G = dL/d(V s)
    gv = (G * V).sum(axis=axes)                      # dL/ds per channel
    d = np.asarray(dora_scale, np.float64).reshape(-1)
    d_dora = gv * multiplier / n
    dn = gv * (-multiplier * d / (n * n))            # dL/dn per channel
    nrm = n - eps
    dV = G * s.reshape(bs) + V * (dn / np.where(nrm > 0, nrm, 1.0)).reshape(bs)
    return dx, dV, d_dora.reshape(np.asarray(dora_scale).shape)
