"""Autograd-level wrappers over the C ABI.

Every adapter algorithm is a "row core": kernels that act on an activation row matrix [M, features]
(``fwd``/``bwd`` below call the C entry points of include/lycoris_amd.h).  Two generic autograd Functions wrap a
core: ``_AdapterLinear`` (nn.Linear: rows = x.view(-1, I)) and ``_AdapterConv2d`` (nn.Conv2d, NCHW: rows = im2col
view, outputs transposed back, col2im accumulating un-rounded fp32 rows in backward).

Factors are fp32 for the kernels (16-bit factors are up-converted -- they are tiny); factor gradients are returned
in the factor's dtype.  ``alpha`` is a Python float (scale * multiplier); a learnable ``scalar`` gate is folded into
the first factor by the caller so its gradient flows through ordinary autograd.
"""
from __future__ import annotations

import torch

from . import _native as N
from . import composite as _host  # the ops' forms for tensors that are not on the HIP device (BASELINE configs[0])

F32_ROWS = 0x100  # LYC_F32_ROWS

# Fused gradient accumulation (optional): when a factor handed to an adapter op is a leaf Parameter whose .grad
# already exists as a contiguous fp32 tensor (e.g. a view of lycoris_amd.grad_sync's arena), the backward kernels
# accumulate straight into it (their outputs are "+=" anyway) and autograd gets None for that input -- no temporary,
# no zero-fill, no separate accumulate kernel.  `_ACCUM["callback"]` is told which parameters were updated so a
# gradient-sync object can count them like a post-accumulate hook would.
_ACCUM = {"enabled": False, "callback": None, "batch_callback": None}

# Host dispatch of the public entry points below:
#   "cpp"    (default) torch.ops.lycoris_amd.* -- TORCH_LIBRARY custom ops, dispatch + autograd in C++ (csrc/torch_ops.cpp)
#   "python" the ctypes-backed torch.autograd.Function classes of this file (same kernels; kept as the readable
#            statement of the host logic and as a cross-check in the tests)
_DISPATCH = {"mode": "cpp", "ext": None}


def set_dispatch(mode: str):
    if mode not in ("cpp", "python"):
        raise ValueError(mode)
    _DISPATCH["mode"] = mode


_OPS = {}


def _cpp() -> bool:
    if _DISPATCH["mode"] != "cpp":
        return False
    if _DISPATCH["ext"] is None:
        ext = N.load_torch_ops()  # raises NativeLibraryError when the extension is missing: no silent fallback
        ext.set_accum(_ACCUM["enabled"], _ACCUM["callback"], _ACCUM["batch_callback"])
        _DISPATCH["ext"] = ext
        # A step boundary makes every cached LoKr operand plane stale, whatever the optimizer does to the version counters
        # (Prodigy / DAdaptation / raw 8-bit kernels write through `p.data`, which does not bump them): every
        # torch.optim.Optimizer.step() marks the cache dirty, and so does the end of every backward pass (csrc/torch_ops.cpp).
        try:
            from torch.optim.optimizer import register_optimizer_step_post_hook
            register_optimizer_step_post_hook(lambda opt, args, kwargs: ext.mark_planes_dirty())
        except ImportError:  # very old torch: the end-of-backward mark alone
            pass
        ns = torch.ops.lycoris_amd  # the resolved overloads: skips the packet's per-call overload resolution (~1 us per call)
        for name in ("lokr_linear", "lokr_adapted_linear", "lokr_linear_group", "lokr_linear_lr_group", "locon_linear_group", "lokr_linear_lr", "lokr_linear_lr2", "locon_linear", "loha_linear", "chan_affine", "lokr_conv2d", "locon_conv2d", "adapter_conv2d", "lokr_conv2d_lr"):
            _OPS[name] = getattr(ns, name).default
    return True


def fused_grad_accumulation(enabled: bool = True, callback=None, batch_callback=None):
    """`callback(param)` is told about every parameter whose gradient has just been accumulated into `.grad` by the kernels;
    `batch_callback(list_of_params)` (optional) takes the reports of a whole grouped weight-gradient flush in ONE call.

    A parameter used by several layer calls of one forward pass is reported once, after the last of its backward nodes: the forward
    counts the nodes it creates (csrc/torch_ops.cpp expect()).  Counts of forward passes that are never differentiated (a validation
    forward outside no_grad, a pass that raised) stay behind: owners call `reset_use_counts()` at their step boundary --
    AdapterGradSync does in zero_grad() / finish().  The recomputation of non-reentrant activation checkpointing is not counted."""
    _ACCUM["enabled"] = bool(enabled)
    _ACCUM["callback"] = callback
    _ACCUM["batch_callback"] = batch_callback if callback is not None else None
    if _DISPATCH["ext"] is not None:
        _DISPATCH["ext"].set_accum(bool(enabled), callback, _ACCUM["batch_callback"])


def locon_reg_staged(enabled: bool = True) -> bool:
    """A/B and regression-test switch: keep the rank-r (LoCon) reduce / expand launches on the register-staged kernel of rounds 1-5
    (csrc/lowrank.h: bneck_kernel) instead of the LDS-DMA kernel of round 6 (csrc/lowrank4.h: bneck4_kernel, and the fused sibling
    sum bneck4_sum_kernel).  Returns the previous setting.  `LYC_BNECK_REG` in the dtype argument of the C ABI."""
    _cpp()
    return bool(_DISPATCH["ext"].locon_reg_staged(bool(enabled)))


def fused_reports(param) -> bool:
    """True when the kernels themselves report `param` to the fused-accumulation callback in the running backward pass (its
    autograd post-accumulate hook, which fires on the undefined gradient the backward node returned, must then stay silent)."""
    if not _cpp():
        return False
    return bool(_DISPATCH["ext"].fused_reports(param))


def deferred_weight_gradients(enabled: bool = True, flush_at: int = 48):
    """LoKr layers whose factor gradients are accumulated straight into `.grad` (fused_grad_accumulation) run only their
    dx launch inside the backward pass; dW1 / dW2 of up to `flush_at` parked layers are then computed by ONE grouped launch
    per 24 layers (lyc_lokr_wgrad_group) -- at the latest when the autograd engine finishes the pass, so `.grad` is complete
    when `backward()` returns.  On by default; `enabled=False` restores one weight-gradient launch per layer."""
    if not _cpp():
        raise RuntimeError("deferred weight gradients live in the C++ dispatch (ops.set_dispatch('cpp'))")
    _DISPATCH["ext"].set_defer(bool(enabled), int(flush_at))


def flush_deferred():
    """Enqueue the weight gradients of every parked layer now (the engine does this by itself at the end of a backward pass)."""
    if _DISPATCH["ext"] is not None:
        _DISPATCH["ext"].flush_deferred()


def discard_deferred() -> int:
    """After a backward pass that raised: drop the parked layers so that they are not added to the next step."""
    return _DISPATCH["ext"].discard_deferred() if _DISPATCH["ext"] is not None else 0


def refresh_lokr_planes(force: bool = False):
    """Repack the cached operand planes -- LoKr's packed w2 planes (csrc/torch_ops.cpp `planes_for`) and, since round 6, LoHa's dW planes
    (`loha_plane_for`) -- of every parameter that changed since they were
    written -- `force`: of every cached parameter -- in grouped launches on the current stream.  Eager training never needs to
    call this (the first layer call after optimizer.step() does it); a caller that REPLAYS captured graphs must capture this
    call (with force=True) in front of the forward pass, so that each replay sees the parameters of its own step."""
    if _cpp():
        _DISPATCH["ext"].refresh_planes(bool(force))


def mark_planes_dirty():
    """Tell the LoKr operand-plane cache that parameters have changed (a step boundary).  torch.optim optimizers and the end of every
    backward pass that ran a LoKr op do this by themselves; an optimizer that is NOT a torch.optim.Optimizer and writes through
    `p.data` / views of a flat arena (no version counter moves) calls this after its update."""
    if _cpp():
        _DISPATCH["ext"].mark_planes_dirty()


def lokr_planes_cache(enabled: bool = True):
    """A/B switch: False drops the cache; the kernels then convert the fp32 factor tile per workgroup (rounds 1-2)."""
    if _cpp():
        _DISPATCH["ext"].set_planes_cache(bool(enabled))


def reset_use_counts():
    """Forget the per-parameter count of pending accumulations (csrc/torch_ops.cpp `expect()` / `notify()`): a forward pass whose
    backward never ran leaves a count behind, which would swallow that parameter's next report.  AdapterGradSync calls this once
    per optimizer step (zero_grad / finish)."""
    if _DISPATCH["ext"] is not None:
        _DISPATCH["ext"].reset_use_counts()


def _grad_targets(factors, needs):
    """Per factor: the tensor the kernel should accumulate into (existing .grad or a fresh zero buffer) and whether
    the result has to be handed back to autograd."""
    bufs, hand_back = [], []
    for t, need in zip(factors, needs):
        if not need:
            bufs.append(None)
            hand_back.append(False)
            continue
        g = t.grad if (_ACCUM["enabled"] and t.is_leaf) else None
        if g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == t.device:
            bufs.append(g)
            hand_back.append(False)
        else:
            bufs.append(torch.zeros(t.shape, dtype=torch.float32, device=t.device))
            hand_back.append(True)
    return bufs, hand_back


def _finish_grads(factors, bufs, hand_back):
    out = []
    for t, b, hb in zip(factors, bufs, hand_back):
        if hb:
            out.append(b.to(t.dtype))
        else:
            out.append(None)
            if b is not None and _ACCUM["callback"] is not None:
                _ACCUM["callback"](t)
    return out


def _amp(x: torch.Tensor) -> torch.Tensor:
    """torch.autocast parity with the reference: its F.linear / F.conv2d run in the autocast dtype (modules/locon.py:
    321-331 under sd-scripts mixed precision: fp32 LayerNorm output in, bf16 delta out).  The native ops take the
    activation dtype from x, so an fp32 x is cast here (differentiably: dx comes back in fp32) -- the adapter then runs
    the 16-bit fast path and `base + delta` stays in the autocast dtype.  Factors are left in fp32."""
    if x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled("cuda"):
        return x.to(torch.get_autocast_dtype("cuda"))
    return x


def _f32c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ---------------------------------------------------------------------------------------------------------------
# row cores
# ---------------------------------------------------------------------------------------------------------------
class _LokrCore:
    """factors = (w1:[a,b], w2:[c,d]);  y = rows @ (kron(w1, w2) * alpha)^T"""
    n_factors = 2

    @staticmethod
    def dims(fs):
        (a, b), (c, d) = fs[0].shape, fs[1].shape
        return b * d, a * c

    @staticmethod
    def fwd(rows, fs, alpha):
        (a, b), (c, d) = fs[0].shape, fs[1].shape
        y = torch.empty((rows.shape[0], a * c), dtype=rows.dtype, device=rows.device)
        N.call("lyc_lokr_linear_fwd", N.ptr(rows), N.ptr(fs[0]), N.ptr(fs[1]), None, N.ptr(y), rows.shape[0], a, b, c, d,
               alpha, N.dtype_code(rows.dtype), N.stream_ptr(rows.device))
        return y, ()

    @staticmethod
    def bwd(g, rows, fs, saved, alpha, need_x, need_f, f32_rows, bufs):
        (a, b), (c, d) = fs[0].shape, fs[1].shape
        want_dx = need_x or need_f[0]  # the w1 gradient shares the pass over g that produces dx
        code = N.dtype_code(rows.dtype) | (F32_ROWS if f32_rows else 0)
        dx = None
        if want_dx:
            dx = torch.empty(rows.shape, dtype=torch.float32 if f32_rows else rows.dtype, device=rows.device)
        dw1, dw2 = bufs
        ws = None
        if dw1 is not None:  # scratch for the per-workgroup w1-gradient partials (caching allocator: no sync, no memset)
            nbytes = int(N.load().lyc_lokr_bwd_workspace_bytes(rows.shape[0], a, b, c, d, code & 0xff))
            if nbytes:
                ws = torch.empty(nbytes, dtype=torch.uint8, device=rows.device)
        N.call("lyc_lokr_linear_bwd", N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(dx), N.ptr(dw1),
               N.ptr(dw2), N.ptr(ws), rows.shape[0], a, b, c, d, alpha, code, N.stream_ptr(rows.device))
        return (dx if need_x else None)


class _LoconCore:
    """factors = (down:[r,I], up:[O,r]);  y = alpha * (rows @ down^T) @ up^T, rank-r intermediate kept in fp32"""
    n_factors = 2

    @staticmethod
    def dims(fs):
        return fs[0].shape[1], fs[1].shape[0]

    @staticmethod
    def fwd(rows, fs, alpha):
        r, I = fs[0].shape
        O = fs[1].shape[0]
        M = rows.shape[0]
        t = torch.empty((M, r), dtype=torch.float32, device=rows.device)  # written by the call
        y = torch.empty((M, O), dtype=rows.dtype, device=rows.device)
        N.call("lyc_locon_linear_fwd", N.ptr(rows), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(t), N.ptr(y), M, I, O, r,
               alpha, N.dtype_code(rows.dtype), N.stream_ptr(rows.device))
        return y, (t,)

    @staticmethod
    def bwd(g, rows, fs, saved, alpha, need_x, need_f, f32_rows, bufs):
        r, I = fs[0].shape
        O = fs[1].shape[0]
        M = rows.shape[0]
        code = N.dtype_code(rows.dtype) | (F32_ROWS if f32_rows else 0)
        dt = torch.empty((M, r), dtype=torch.float32, device=rows.device)  # scratch, written by the call
        dx = torch.empty(rows.shape, dtype=torch.float32 if f32_rows else rows.dtype, device=rows.device) if need_x else None
        dd, du = bufs
        N.call("lyc_locon_linear_bwd", N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(saved[0]), N.ptr(dt),
               N.ptr(dx), N.ptr(dd), N.ptr(du), M, I, O, r, alpha, code, N.stream_ptr(rows.device))
        return dx


class _LohaCore:
    """factors = (w1a:[O,r], w1b:[r,I], w2a:[O,r], w2b:[r,I]);  y = rows @ (((w1a w1b) * (w2a w2b)) * alpha)^T.
    The dense dW operand images live in a scratch buffer written by the forward kernels and re-read by backward."""
    n_factors = 4

    @staticmethod
    def dims(fs):
        return fs[1].shape[1], fs[0].shape[0]

    @staticmethod
    def fwd(rows, fs, alpha):
        O, r = fs[0].shape
        I = fs[1].shape[1]
        M = rows.shape[0]
        code = N.dtype_code(rows.dtype)
        ws = torch.empty(int(N.load().lyc_loha_workspace_bytes(O, I, code)), dtype=torch.uint8, device=rows.device)
        y = torch.empty((M, O), dtype=rows.dtype, device=rows.device)
        N.call("lyc_loha_linear_fwd", N.ptr(rows), *[N.ptr(t) for t in fs], N.ptr(ws), N.ptr(y), M, I, O, r, alpha,
               code, N.stream_ptr(rows.device))
        return y, (ws,)

    @staticmethod
    def bwd(g, rows, fs, saved, alpha, need_x, need_f, f32_rows, bufs):
        O, r = fs[0].shape
        I = fs[1].shape[1]
        M = rows.shape[0]
        code = N.dtype_code(rows.dtype) | (F32_ROWS if f32_rows else 0)
        any_f = any(need_f)
        dx = torch.empty(rows.shape, dtype=torch.float32 if f32_rows else rows.dtype, device=rows.device) if need_x else None
        grads = list(bufs)
        if any_f and not all(need_f):  # the factor-gradient kernel produces the four gradients as a set
            grads = [b if b is not None else torch.zeros_like(t) for b, t in zip(bufs, fs)]
        gw = torch.empty((O, I), dtype=torch.float32, device=rows.device) if any_f else None
        N.call("lyc_loha_linear_bwd", N.ptr(g), N.ptr(rows), *[N.ptr(t) for t in fs], N.ptr(saved[0]), N.ptr(gw),
               N.ptr(dx), *[N.ptr(t) for t in grads], M, I, O, r, alpha, code, N.stream_ptr(rows.device))
        return dx


# ---------------------------------------------------------------------------------------------------------------
# Conv2d lowering helpers (no autograd: used inside the Functions)
# ---------------------------------------------------------------------------------------------------------------
def _conv_out(H, W, k, s, p, d):
    Ho = (H + 2 * p[0] - d[0] * (k[0] - 1) - 1) // s[0] + 1
    Wo = (W + 2 * p[1] - d[1] * (k[1] - 1) - 1) // s[1] + 1
    return Ho, Wo


def _is_pointwise(geom):
    k, s, p, d = geom
    return k == (1, 1) and s == (1, 1) and p == (0, 0)


def _to_rows(t):  # [B, C, *sp] -> [B*P, C]
    B, C = t.shape[0], t.shape[1]
    P = t[0, 0].numel()
    rows = torch.empty((B * P, C), dtype=t.dtype, device=t.device)
    N.call("lyc_nchw_to_rows", N.ptr(t), N.ptr(rows), B, C, P, N.dtype_code(t.dtype), N.stream_ptr(t.device))
    return rows


def _from_rows(rows, B, spatial):  # [B*P, C] -> [B, C, *sp]
    C = rows.shape[1]
    out = torch.empty((B, C, *spatial), dtype=rows.dtype, device=rows.device)
    N.call("lyc_rows_to_nchw", N.ptr(rows), N.ptr(out), B, C, out[0, 0].numel(), N.dtype_code(rows.dtype),
           N.stream_ptr(rows.device))
    return out


def _im2col(x, geom):
    k, s, p, d = geom
    B, C, H, W = x.shape
    Ho, Wo = _conv_out(H, W, k, s, p, d)
    cols = torch.empty((B * Ho * Wo, C * k[0] * k[1]), dtype=x.dtype, device=x.device)
    N.call("lyc_im2col", N.ptr(x), N.ptr(cols), B, C, H, W, k[0], k[1], s[0], s[1], p[0], p[1], d[0], d[1],
           N.dtype_code(x.dtype), N.stream_ptr(x.device))
    return cols


def _col2im(dcols, xshape, dtype, geom):
    k, s, p, d = geom
    B, C, H, W = xshape
    dx = torch.empty(xshape, dtype=dtype, device=dcols.device)
    code = N.dtype_code(dtype) | (F32_ROWS if dcols.dtype == torch.float32 and dtype != torch.float32 else 0)
    N.call("lyc_col2im", N.ptr(dcols), N.ptr(dx), B, C, H, W, k[0], k[1], s[0], s[1], p[0], p[1], d[0], d[1], code,
           N.stream_ptr(dcols.device))
    return dx


# round 6: LoHa on a 16-bit Conv2d with C % 8 == 0 -- im2col / col2im on the NHWC row matrix with window-major columns
# (lyc_im2col_rows / lyc_col2im_rows: 16-byte vectors), factors with their columns permuted the same way (csrc/torch_ops.cpp)
def _window_major(core, x, geom):
    return core is _LohaCore and x.dtype != torch.float32 and x.shape[1] % 8 == 0 and not _is_pointwise(geom)


def _to_window_major(f, C, kk):
    r = f.shape[0]
    return _f32c(f.detach()).view(r, C, kk).transpose(1, 2).contiguous().view(r, kk * C)


def _im2col_rows(x_rows, xshape, geom):
    k, s, p, d = geom
    B, C, H, W = xshape
    Ho, Wo = _conv_out(H, W, k, s, p, d)
    cols = torch.empty((B * Ho * Wo, k[0] * k[1] * C), dtype=x_rows.dtype, device=x_rows.device)
    N.call("lyc_im2col_rows", N.ptr(x_rows), N.ptr(cols), B, C, H, W, k[0], k[1], s[0], s[1], p[0], p[1], d[0], d[1],
           N.dtype_code(x_rows.dtype), N.stream_ptr(x_rows.device))
    return cols


def _col2im_rows(dcols, xshape, dtype, geom):
    k, s, p, d = geom
    B, C, H, W = xshape
    dx_rows = torch.empty((B * H * W, C), dtype=dtype, device=dcols.device)
    code = N.dtype_code(dtype) | (F32_ROWS if dcols.dtype == torch.float32 else 0)
    N.call("lyc_col2im_rows", N.ptr(dcols), N.ptr(dx_rows), B, C, H, W, k[0], k[1], s[0], s[1], p[0], p[1], d[0], d[1], code,
           N.stream_ptr(dcols.device))
    return dx_rows


# ---------------------------------------------------------------------------------------------------------------
# generic autograd Functions
# ---------------------------------------------------------------------------------------------------------------
class _AdapterLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, core, alpha, x, *factors):
        N.require_device(x, "input")
        fs = [_f32c(t) for t in factors]
        I, O = core.dims(fs)
        if x.shape[-1] != I:
            raise ValueError(f"adapter expects {I} input features, got {tuple(x.shape)}")
        rows = x.reshape(-1, I)
        rows = rows if rows.is_contiguous() else rows.contiguous()
        y, saved = core.fwd(rows, fs, float(alpha))
        ctx.save_for_backward(rows, *factors, *saved)
        ctx.meta = (core, float(alpha), x.shape, len(factors))
        return y.view(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, g):
        core, alpha, xshape, nf = ctx.meta
        rows, factors, saved = ctx.saved_tensors[0], ctx.saved_tensors[1:1 + nf], ctx.saved_tensors[1 + nf:]
        fs = [_f32c(t) for t in factors]
        g2 = g.reshape(-1, g.shape[-1])
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        need_x, need_f = ctx.needs_input_grad[2], list(ctx.needs_input_grad[3:])
        bufs, hand_back = _grad_targets(factors, need_f)
        dx = core.bwd(g2, rows, fs, saved, alpha, need_x, need_f, False, bufs)
        return (None, None, dx.view(xshape) if need_x else None, *_finish_grads(factors, bufs, hand_back))


class _AdapterConv2d(torch.autograd.Function):
    """Conv2d (NCHW, groups=1) form of a row core.  factors are the 2-D views ([.., I*kh*kw]) of the conv factors."""

    @staticmethod
    def forward(ctx, core, alpha, geom, x, *factors):
        N.require_device(x, "input")
        if x.dim() != 4:
            raise ValueError(f"Conv2d adapter expects NCHW input, got shape {tuple(x.shape)}")
        fs = [_f32c(t) for t in factors]
        k, s, p, d = geom
        B, C, H, W = x.shape
        I, O = core.dims(fs)
        if I != C * k[0] * k[1]:
            raise ValueError(f"adapter expects {I} = C*kh*kw im2col features, input has C={C}, kernel={k}")
        x_cl = False
        wm = _window_major(core, x, geom)
        if _is_pointwise(geom):  # a 1x1 conv IS the row op on the NHWC pixel rows: free for a channels_last tensor
            rows, copied = _rows_view(x)
            x_cl = not copied
        elif wm:
            xr, copied = _rows_view(x)
            x_cl = not copied
            rows = _im2col_rows(xr, x.shape, geom)
            kk = k[0] * k[1]
            fs = [fs[0], _to_window_major(fs[1], C, kk), fs[2], _to_window_major(fs[3], C, kk)]
        else:
            rows = _im2col(x.contiguous(), geom)
        y_rows, saved = core.fwd(rows, fs, float(alpha))
        ctx.save_for_backward(rows, *factors, *saved)
        ctx.meta = (core, float(alpha), geom, x.shape, len(factors), x_cl, wm)
        sp = _conv_out(H, W, k, s, p, d)
        if x_cl:  # keep the caller's memory format: the row matrix is the channels_last tensor
            return y_rows.view(B, sp[0], sp[1], O).permute(0, 3, 1, 2)
        return _from_rows(y_rows, B, sp)

    @staticmethod
    def backward(ctx, g):
        core, alpha, geom, xshape, nf, x_cl, wm = ctx.meta
        rows, factors, saved = ctx.saved_tensors[0], ctx.saved_tensors[1:1 + nf], ctx.saved_tensors[1 + nf:]
        fs = [_f32c(t) for t in factors]
        g_rows, _ = _rows_view(g)
        need_x, need_f = ctx.needs_input_grad[3], list(ctx.needs_input_grad[4:])
        pointwise = _is_pointwise(geom)
        # k > 1: col2im sums up to kh*kw row entries per pixel -> keep them in fp32 and round once
        bufs, hand_back = _grad_targets(factors, need_f)
        if wm:  # the b-side factors and their gradients in the window-major layout; un-permuted into the callers' buffers
            C, kk, r = xshape[1], geom[0][0] * geom[0][1], fs[1].shape[0]
            fs = [fs[0], _to_window_major(fs[1], C, kk), fs[2], _to_window_major(fs[3], C, kk)]
            wbufs = list(bufs)
            for i in (1, 3):
                if bufs[i] is not None:
                    wbufs[i] = torch.zeros((r, kk * C), dtype=torch.float32, device=g.device)
            dx_rows = core.bwd(g_rows, rows, fs, saved, alpha, need_x, need_f, True, wbufs)
            for i in (1, 3):
                if bufs[i] is not None:
                    bufs[i].view(r, C, kk).add_(wbufs[i].view(r, kk, C).transpose(1, 2))
            dx = None
            if need_x:
                dxr = _col2im_rows(dx_rows, xshape, rows.dtype, geom)
                dx = (dxr.view(xshape[0], xshape[2], xshape[3], xshape[1]).permute(0, 3, 1, 2) if x_cl
                      else _from_rows(dxr, xshape[0], xshape[2:]))
            return (None, None, None, dx, *_finish_grads(factors, bufs, hand_back))
        dx_rows = core.bwd(g_rows, rows, fs, saved, alpha, need_x, need_f, not pointwise, bufs)
        dx = None
        if need_x and pointwise and x_cl:
            dx = dx_rows.view(xshape[0], xshape[2], xshape[3], xshape[1]).permute(0, 3, 1, 2)
        elif need_x:
            dx = _from_rows(dx_rows, xshape[0], xshape[2:]) if pointwise else _col2im(dx_rows, xshape, rows.dtype, geom)
        return (None, None, None, dx, *_finish_grads(factors, bufs, hand_back))


# ---------------------------------------------------------------------------------------------------------------
# LoKr Conv2d: implicit GEMM on NHWC rows (no im2col)
# ---------------------------------------------------------------------------------------------------------------
def _rows_view(t):
    """[B, C, H, W] -> ([B*H*W, C] NHWC row matrix, made_copy).  A channels_last tensor already is one."""
    B, C, H, W = t.shape
    if t.is_contiguous(memory_format=torch.channels_last) and not (C == 1 or H * W == 1):
        return t.permute(0, 2, 3, 1).reshape(B * H * W, C), False
    return _to_rows(t.contiguous()), True


def _lokr_conv_implicit_ok(x, w1, w2):
    a, b = w1.shape
    c, d = w2.shape[0], w2.shape[1]
    return (x.dtype in (torch.bfloat16, torch.float16) and a == b and a in (4, 8, 16) and c % 8 == 0 and d % 8 == 0
            and x.shape[2] * x.shape[3] < (1 << 30))


class _LokrConv2dImplicit(torch.autograd.Function):
    """LoKr on nn.Conv2d without materialising im2col: lyc_lokr_conv2d_fwd / _bwd (include/lycoris_amd.h).
    w2 is the reference's [c, d, kh, kw] parameter; the kernels want [c, kh*kw, d] (window in front of the channel
    index), which is a free view when the parameter lives in channels_last memory format and one small copy otherwise."""

    @staticmethod
    def forward(ctx, alpha, geom, x, w1, w2):
        N.require_device(x, "input")
        k, s, p, d_ = geom
        B, C, H, W = x.shape
        a, b = w1.shape
        c, d = w2.shape[0], w2.shape[1]
        if C != b * d:
            raise ValueError(f"adapter expects {b * d} input channels, got {tuple(x.shape)}")
        rows, copied = _rows_view(x)
        w1f = _f32c(w1)
        w2p = _f32c(w2.detach().permute(0, 2, 3, 1))  # [c, kh, kw, d]; no copy for channels_last fp32 parameters
        Ho, Wo = _conv_out(H, W, k, s, p, d_)
        y_rows = torch.empty((B * Ho * Wo, a * c), dtype=x.dtype, device=x.device)
        N.call("lyc_lokr_conv2d_fwd", N.ptr(rows), N.ptr(w1f), N.ptr(w2p), N.ptr(y_rows), B, H, W, a, b, c, d, k[0], k[1],
               s[0], s[1], p[0], p[1], d_[0], d_[1], float(alpha), N.dtype_code(x.dtype), N.stream_ptr(x.device))
        ctx.save_for_backward(rows, w1, w2)
        ctx.meta = (float(alpha), geom, x.shape, (Ho, Wo), not copied)
        if not copied:  # channels_last in -> channels_last out, no transposes at all
            return y_rows.view(B, Ho, Wo, a * c).permute(0, 3, 1, 2)
        return _from_rows(y_rows, B, (Ho, Wo))

    @staticmethod
    def backward(ctx, g):
        alpha, geom, xshape, (Ho, Wo), x_cl = ctx.meta
        rows, w1, w2 = ctx.saved_tensors
        k, s, p, d_ = geom
        B, C, H, W = xshape
        a, b = w1.shape
        c, d = w2.shape[0], w2.shape[1]
        g_rows, _ = _rows_view(g)
        need_x, need_w1, need_w2 = ctx.needs_input_grad[2], ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        w1f = _f32c(w1)
        w2p = _f32c(w2.detach().permute(0, 2, 3, 1))
        code = N.dtype_code(rows.dtype)
        dx_rows = torch.empty((B * H * W, C), dtype=rows.dtype, device=rows.device) if (need_x or need_w1) else None
        (dw1,), hb1 = _grad_targets([w1], [need_w1])
        # dw2 in the kernels' [c, kh, kw, d] layout: straight into w2.grad when that has the same memory layout
        dw2p, hb2 = None, False
        if need_w2:
            gr = w2.grad if (_ACCUM["enabled"] and w2.is_leaf) else None
            if (gr is not None and gr.dtype == torch.float32 and gr.device == w2.device
                    and gr.permute(0, 2, 3, 1).is_contiguous()):
                dw2p = gr.permute(0, 2, 3, 1)
            else:
                dw2p, hb2 = torch.zeros((c, k[0], k[1], d), dtype=torch.float32, device=rows.device), True
        ws = None
        if dw1 is not None:
            nbytes = int(N.load().lyc_lokr_conv2d_bwd_workspace_bytes(B, H, W, a, b, d))
            if nbytes:
                ws = torch.empty(nbytes, dtype=torch.uint8, device=rows.device)
        # stride 1: a [kh, kw, c, d] copy of the (small) factor lets the transposed convolution use full K segments
        w2t = w2.detach().float().permute(2, 3, 0, 1).contiguous() if (dx_rows is not None and s == (1, 1)) else None
        N.call("lyc_lokr_conv2d_bwd", N.ptr(g_rows), N.ptr(rows), N.ptr(w1f), N.ptr(w2p), N.ptr(w2t), N.ptr(dx_rows), N.ptr(dw1),
               N.ptr(dw2p), N.ptr(ws), B, H, W, a, b, c, d, k[0], k[1], s[0], s[1], p[0], p[1], d_[0], d_[1], alpha, code,
               N.stream_ptr(rows.device))
        dx = None
        if need_x:
            dx = dx_rows.view(B, H, W, C).permute(0, 3, 1, 2) if x_cl else _from_rows(dx_rows, B, (H, W))
        gw1 = _finish_grads([w1], [dw1], hb1)[0]
        gw2 = None
        if need_w2:
            if hb2:
                gw2 = dw2p.permute(0, 3, 1, 2).to(w2.dtype)
            elif _ACCUM["callback"] is not None:
                _ACCUM["callback"](w2)
        return None, None, dx, gw1, gw2


def _locon_conv_implicit_ok(x, down, up):
    r, C = down.shape[0], down.shape[1]
    taps = down.shape[2] * down.shape[3]
    return (x.dtype in (torch.bfloat16, torch.float16) and C % 16 == 0 and up.shape[0] % 8 == 0 and r % 4 == 0 and r <= 16
            and taps <= 64 and taps * r <= 144 and x.shape[2] * x.shape[3] < (1 << 30))


def _cl_grad_target(p, need, shape_p):
    """Gradient buffer for a 4-D parameter whose kernels work on the permute(0, 2, 3, 1) (window-major) layout:
    p.grad itself when it is fp32 and lives in that layout (channels_last), else a fresh zero buffer to hand back."""
    if not need:
        return None, False
    gr = p.grad if (_ACCUM["enabled"] and p.is_leaf) else None
    if (gr is not None and gr.dtype == torch.float32 and gr.device == p.device and gr.permute(0, 2, 3, 1).is_contiguous()):
        return gr.permute(0, 2, 3, 1), False
    return torch.zeros(shape_p, dtype=torch.float32, device=p.device), True


class _LoconConv2dImplicit(torch.autograd.Function):
    """LoCon on nn.Conv2d without materialising im2col / col2im: lyc_locon_conv2d_fwd / _bwd (include/lycoris_amd.h).
    down is the reference's lora_down.weight [r, C, kh, kw] (the kernels read it as [r, kh, kw, C]: a free view of a
    channels_last parameter, one small copy otherwise), up is lora_up.weight [O, r, 1, 1]."""

    @staticmethod
    def forward(ctx, alpha, geom, x, down, up):
        N.require_device(x, "input")
        k, s, p, d_ = geom
        B, C, H, W = x.shape
        r, O = down.shape[0], up.shape[0]
        if C != down.shape[1]:
            raise ValueError(f"adapter expects {down.shape[1]} input channels, got {tuple(x.shape)}")
        rows, copied = _rows_view(x)
        down_p = _f32c(down.detach().permute(0, 2, 3, 1))
        up2 = _f32c(up.detach().reshape(O, r))
        Ho, Wo = _conv_out(H, W, k, s, p, d_)
        t = torch.empty((B * Ho * Wo, r), dtype=torch.float32, device=x.device)
        y_rows = torch.empty((B * Ho * Wo, O), dtype=x.dtype, device=x.device)
        N.call("lyc_locon_conv2d_fwd", N.ptr(rows), N.ptr(down_p), N.ptr(up2), N.ptr(t), N.ptr(y_rows), B, H, W, C, O, r,
               k[0], k[1], s[0], s[1], p[0], p[1], d_[0], d_[1], float(alpha), N.dtype_code(x.dtype), N.stream_ptr(x.device))
        ctx.save_for_backward(rows, down, up, t)
        ctx.meta = (float(alpha), geom, x.shape, (Ho, Wo), not copied)
        if not copied:  # channels_last in -> channels_last out
            return y_rows.view(B, Ho, Wo, O).permute(0, 3, 1, 2)
        return _from_rows(y_rows, B, (Ho, Wo))

    @staticmethod
    def backward(ctx, g):
        alpha, geom, xshape, (Ho, Wo), x_cl = ctx.meta
        rows, down, up, t = ctx.saved_tensors
        k, s, p, d_ = geom
        B, C, H, W = xshape
        r, O = down.shape[0], up.shape[0]
        g_rows, _ = _rows_view(g)
        need_x, need_d, need_u = ctx.needs_input_grad[2], ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        down_p = _f32c(down.detach().permute(0, 2, 3, 1))
        up2 = _f32c(up.detach().reshape(O, r))
        dt = torch.empty((B * Ho * Wo, r), dtype=torch.float32, device=rows.device)
        dx_rows = torch.empty((B * H * W, C), dtype=rows.dtype, device=rows.device) if need_x else None
        ddp, hbd = _cl_grad_target(down, need_d, (r, k[0], k[1], C))
        (du,), hbu = _grad_targets([up], [need_u])
        N.call("lyc_locon_conv2d_bwd", N.ptr(g_rows), N.ptr(rows), N.ptr(down_p), N.ptr(up2), N.ptr(t), N.ptr(dt),
               N.ptr(dx_rows), N.ptr(ddp), N.ptr(du), B, H, W, C, O, r, k[0], k[1], s[0], s[1], p[0], p[1], d_[0], d_[1],
               alpha, N.dtype_code(rows.dtype), N.stream_ptr(rows.device))
        dx = None
        if need_x:
            dx = dx_rows.view(B, H, W, C).permute(0, 3, 1, 2) if x_cl else _from_rows(dx_rows, B, (H, W))
        gd = None
        if need_d:
            if hbd:
                gd = ddp.permute(0, 3, 1, 2).to(down.dtype)
            elif _ACCUM["callback"] is not None:
                _ACCUM["callback"](down)
        gu = _finish_grads([up], [du], hbu)[0]
        return None, None, dx, gd, gu


# ---------------------------------------------------------------------------------------------------------------
# (IA)^3 per-channel affine
# ---------------------------------------------------------------------------------------------------------------
def _chan_dims(t: torch.Tensor, chan_dim: int):
    C = t.shape[chan_dim]
    outer = 1
    for s in t.shape[:chan_dim]:
        outer *= s
    inner = 1
    for s in t.shape[chan_dim + 1:]:
        inner *= s
    return outer, C, inner


class _ChanAffine(torch.autograd.Function):
    """out = a * (s0 + w[c]*mult) - bias[c]*w[c]*mult   over the channel dimension ``chan_dim``.

    s0 = 1, bias = layer bias : (IA)^3 out-side  y = base + (base - bias) * w*mult
    s0 = 0, bias = None       : (IA)^3 in-side   x * (w*mult)
    """

    @staticmethod
    def forward(ctx, a, w, bias, s0, mult, chan_dim):
        N.require_device(a, "input")
        a = a.contiguous()
        chan_dim = chan_dim % a.dim()
        outer, C, inner = _chan_dims(a, chan_dim)
        wf = _f32c(w).reshape(-1)
        if wf.numel() != C:
            raise ValueError(f"(IA)^3 weight has {wf.numel()} entries, channel dim has {C}")
        bf = None if bias is None else _f32c(bias).reshape(-1)
        out = torch.empty_like(a)
        N.call("lyc_chan_scale", N.ptr(a), N.ptr(wf), N.ptr(bf), N.ptr(out), outer, C, inner, float(s0),
               float(mult), N.dtype_code(a.dtype), N.stream_ptr(a.device))
        ctx.save_for_backward(a, w, bias)
        ctx.cfg = (float(s0), float(mult), chan_dim)
        return out

    @staticmethod
    def backward(ctx, g):
        a, w, bias = ctx.saved_tensors
        s0, mult, chan_dim = ctx.cfg
        g = g.contiguous()
        outer, C, inner = _chan_dims(a, chan_dim)
        wf = _f32c(w).reshape(-1)
        bf = None if bias is None else _f32c(bias).reshape(-1)
        code, st = N.dtype_code(a.dtype), N.stream_ptr(a.device)
        da = dw = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(g)
        dwf = hb = None
        if ctx.needs_input_grad[1]:
            (dwf,), hb = _grad_targets([w], [True])  # the kernel adds into its output: straight into w.grad
        if da is not None or dwf is not None:  # da and dw in ONE pass over g (lyc_chan_bwd, round 6)
            N.call("lyc_chan_bwd", N.ptr(g), N.ptr(a), N.ptr(wf), N.ptr(bf), N.ptr(da), N.ptr(dwf), outer, C, inner, s0, mult, code, st)
        if dwf is not None:
            dw = _finish_grads([w], [dwf], hb)[0]
        return da, dw, None, None, None, None


# ---------------------------------------------------------------------------------------------------------------
# public functional entry points
# ---------------------------------------------------------------------------------------------------------------
def lokr_linear_fusable(x, w1, w2, base):
    """can `base + delta` be formed in the kernel's epilogue?  (the 16-bit fast path of kron3_kernel, same dtype / shape)"""
    a, b = w1.shape
    return (base is not None and base.dtype in (torch.bfloat16, torch.float16) and base.is_contiguous()
            and (x.dtype == base.dtype or (x.dtype == torch.float32 and torch.is_autocast_enabled("cuda")
                                           and torch.get_autocast_dtype("cuda") == base.dtype))
            and a == b and 16 % a == 0 and w2.shape[1] % 8 == 0 and w2.shape[0] % 4 == 0
            and tuple(base.shape) == (*x.shape[:-1], a * w2.shape[0]) and _DISPATCH["mode"] == "cpp")


def lokr_linear(x, w1, w2, alpha=1.0, base=None):
    """w1:[a,b]  w2:[c,d]  x:[..., b*d] -> [..., a*c];  with `base` (the frozen layer's output): base + delta, fused into the
    kernel's epilogue when lokr_linear_fusable(...), a separate add otherwise"""
    if not x.is_cuda:  # device dispatch: host tensors take the ATen composite form (composite.py); device tensors the HIP kernels, always
        return _host.lokr_linear(x, w1, w2, alpha, base)
    if base is not None and not lokr_linear_fusable(x, w1, w2, base):
        return base + lokr_linear(x, w1, w2, alpha)
    if _cpp():
        return _OPS["lokr_linear"](x, w1, w2, float(alpha), base)
    return _AdapterLinear.apply(_LokrCore, alpha, _amp(x), w1, w2)


def lokr_linear_group(x, w1s, w2s, alphas, bases=None):
    """n LoKr projections of ONE input (to_q / to_k / to_v of a self-attention block, to_k / to_v of a cross-attention: all with equal
    factor shapes) as one call: one forward launch, one autograd node with n outputs, one backward dx launch (csrc/torch_ops.cpp
    LokrLinearGroupFn over lyc_lokr_linear_fwd_group / _bwd_group).  Returns the list of `base_i + delta_i` (or `delta_i` without
    bases); bit-identical to n lokr_linear calls.  Host tensors, tracing and the python dispatch run the n calls one by one."""
    n = len(w1s)
    if len(w2s) != n or len(alphas) != n or (bases is not None and len(bases) != n):
        raise ValueError("lokr_linear_group: w1s, w2s, alphas (and bases) must have one entry per problem")
    if n == 1 or not x.is_cuda or not _cpp() or torch.compiler.is_compiling():
        return [lokr_linear(x, w1s[i], w2s[i], alphas[i], None if bases is None else bases[i]) for i in range(n)]
    if bases is not None and not all(lokr_linear_fusable(x, w1s[i], w2s[i], bases[i]) for i in range(n)):
        ys = lokr_linear_group(x, w1s, w2s, alphas)
        return [b + y for b, y in zip(bases, ys)]
    factors = [t for pair in zip(w1s, w2s) for t in pair]
    return list(_OPS["lokr_linear_group"](x, factors, [float(a) for a in alphas], [] if bases is None else list(bases)))


def lokr_linear_ownable(x, w1, w2, weight, bias=None):
    """can ONE autograd node hold the frozen nn.Linear layer and its LoKr adapter (lokr_adapted_linear)?  A frozen 16-bit weight in the
    dtype the activation has (or is autocast to), eager device tensors, the fused `base + delta` epilogue's shape rules."""
    if not (x.is_cuda and _DISPATCH["mode"] == "cpp") or torch.compiler.is_compiling() or torch.is_inference_mode_enabled():
        return False
    if weight.dtype not in (torch.bfloat16, torch.float16) or weight.requires_grad or weight.dim() != 2 or not weight.is_cuda:
        return False
    if bias is not None and (bias.requires_grad or bias.dtype != weight.dtype):
        return False
    act = x.dtype
    if act == torch.float32 and torch.is_autocast_enabled("cuda"):
        act = torch.get_autocast_dtype("cuda")
    a, b = w1.shape
    return (act == weight.dtype and a == b and 16 % a == 0 and w2.shape[1] % 8 == 0 and w2.shape[0] % 4 == 0
            and weight.shape[0] == a * w2.shape[0] and weight.shape[1] == b * w2.shape[1] and x.shape[-1] == weight.shape[1])


def lokr_adapted_linear(x, weights, biases, w1s, w2s, alphas):
    """[x W_i^T + bias_i + (w1_i (x) w2_i) x * alpha_i]: n frozen nn.Linear layers that read ONE tensor, together with their LoKr adapters,
    as ONE autograd node (round 6; reference modules/lokr.py:551-566 `base + delta` per layer).  Forward: the library GEMMs, then the
    adapter launch with the fused `base + delta` epilogue (one grouped launch for n >= 2).  Backward: the adapter's dx (summed in registers
    over the set), then `dx += g_i W_i` through the library GEMM's accumulate epilogue -- the input has one consumer in the graph, so the
    engine's elementwise `dx_base + dx_adapter` pass (and the n - 1 more of a set) does not exist.  The weights must be frozen
    (lokr_linear_ownable); callers fall back to F.linear + lokr_linear_group otherwise."""
    n = len(w1s)
    if len(weights) != n or len(biases) != n or len(w2s) != n or len(alphas) != n:
        raise ValueError("lokr_adapted_linear: weights, biases, w1s, w2s, alphas must have one entry per layer")
    if not all(lokr_linear_ownable(x, w1s[i], w2s[i], weights[i], biases[i]) for i in range(n)) or not _cpp():
        bases = [torch.nn.functional.linear(x, weights[i], biases[i]) for i in range(n)]
        return lokr_linear_group(x, w1s, w2s, alphas, bases)
    factors = [t for pair in zip(w1s, w2s) for t in pair]
    return list(_OPS["lokr_adapted_linear"](x, factors, [float(a) for a in alphas], list(weights), list(biases)))


def lokr_linear_lr_group(x, w1s, w2as, w2bs, alphas, bases=None):
    """lokr_linear_group for the low-rank second factor w2 = w2a @ w2b (reference modules/lokr.py:131-136): the planes of every problem
    are packed from its pair, the weight gradients go through the grouped chain rule -- as lokr_linear_lr, n projections per call."""
    n = len(w1s)
    if len(w2as) != n or len(w2bs) != n or len(alphas) != n or (bases is not None and len(bases) != n):
        raise ValueError("lokr_linear_lr_group: w1s, w2as, w2bs, alphas (and bases) must have one entry per problem")
    if n == 1 or not x.is_cuda or not _cpp() or torch.compiler.is_compiling():
        return [lokr_linear_lr(x, w1s[i], w2as[i], w2bs[i], alphas[i], None if bases is None else bases[i]) for i in range(n)]
    if bases is not None and not all(lokr_linear_fusable(x, w1s[i], _Shape2(w2as[i].shape[0], w2bs[i].shape[1]), bases[i]) for i in range(n)):
        ys = lokr_linear_lr_group(x, w1s, w2as, w2bs, alphas)
        return [b + y for b, y in zip(bases, ys)]
    factors = [t for tri in zip(w1s, w2as, w2bs) for t in tri]
    return list(_OPS["lokr_linear_lr_group"](x, factors, [float(a) for a in alphas], [] if bases is None else list(bases)))


def lokr_linear_lr(x, w1, w2a, w2b, alpha=1.0, base=None):
    """LoKr with a low-rank second factor (reference modules/lokr.py:131-136: w2 = w2_a @ w2_b).  w1:[a,b]  w2a:[c,r]  w2b:[r,d].
    The product is not materialised: the kernels' operand planes are packed from the two factors and the chain rule of the product
    runs in the grouped weight-gradient launch (csrc/kron_conv.h kron_lr_chain_kernel).  Under tracing / python dispatch the product
    is formed by autograd-visible `w2a @ w2b` and handed to lokr_linear."""
    if not x.is_cuda or not _cpp() or torch.compiler.is_compiling():
        return lokr_linear(x, w1, w2a @ w2b, alpha, base)
    if base is not None and not lokr_linear_fusable(x, w1, _Shape2(w2a.shape[0], w2b.shape[1]), base):
        return base + lokr_linear_lr(x, w1, w2a, w2b, alpha)
    return _OPS["lokr_linear_lr"](x, w1, w2a, w2b, float(alpha), base)


def lokr_linear_lr2(x, w1a, w1b, w2a, w2b, alpha=1.0, base=None):
    """LoKr with BOTH factors low-rank (`decompose_both`, reference modules/lokr.py:94-104): w1 = w1a [a, r] @ w1b [r, b],
    w2 = w2a [c, r] @ w2b [r, d].  The small product is formed once per call below autograd, both weight gradients go through the
    grouped chain-rule launch; tracing / python dispatch / CPU form autograd-visible products and call lokr_linear."""
    if not x.is_cuda or not _cpp() or torch.compiler.is_compiling():
        return lokr_linear(x, w1a @ w1b, w2a @ w2b, alpha, base)
    if base is not None and not lokr_linear_fusable(x, _Shape2(w1a.shape[0], w1b.shape[1]), _Shape2(w2a.shape[0], w2b.shape[1]), base):
        return base + lokr_linear_lr2(x, w1a, w1b, w2a, w2b, alpha)
    return _OPS["lokr_linear_lr2"](x, w1a, w1b, w2a, w2b, float(alpha), base)


class _Shape2:
    """stands in for a [c, d] tensor where only `.shape` is read"""

    def __init__(self, c, d):
        self.shape = (c, d)


def locon_linear(x, down, up, alpha=1.0):
    """down:[r,I]  up:[O,r]"""
    if not x.is_cuda:
        return _host.locon_linear(x, down, up, alpha)
    if _cpp():
        return _OPS["locon_linear"](x, down, up, float(alpha))
    return _AdapterLinear.apply(_LoconCore, alpha, _amp(x), down, up)


def locon_linear_group(x, downs, ups, alphas):
    """n LoCon projections of ONE input (to_q / to_k / to_v of an attention block: equal factor shapes) as one call: one forward launch,
    one autograd node with n outputs, one backward dx launch + a one-pass sum (csrc/torch_ops.cpp LoconLinearGroupFn over
    lyc_locon_linear_fwd_group / _bwd_group).  Returns the list of deltas; bit-identical to n locon_linear calls.  Host tensors,
    tracing and the python dispatch run the n calls one by one."""
    n = len(downs)
    if len(ups) != n or len(alphas) != n:
        raise ValueError("locon_linear_group: downs, ups and alphas must have one entry per problem")
    if n == 1 or not x.is_cuda or not _cpp() or torch.compiler.is_compiling():
        return [locon_linear(x, downs[i], ups[i], alphas[i]) for i in range(n)]
    factors = [t for pair in zip(downs, ups) for t in pair]
    return list(_OPS["locon_linear_group"](x, factors, [float(a) for a in alphas]))


def loha_linear(x, w1a, w1b, w2a, w2b, alpha=1.0):
    """w*a:[O,r]  w*b:[r,I]"""
    if not x.is_cuda:
        return _host.loha_linear(x, w1a, w1b, w2a, w2b, alpha)
    if _cpp():
        return _OPS["loha_linear"](x, w1a, w1b, w2a, w2b, float(alpha))
    return _AdapterLinear.apply(_LohaCore, alpha, _amp(x), w1a, w1b, w2a, w2b)


def chan_affine(a, w, bias=None, s0=0.0, mult=1.0, chan_dim=-1):
    if not a.is_cuda:
        return _host.chan_affine(a, w, bias, s0, mult, chan_dim)
    if _cpp():
        return _OPS["chan_affine"](a, w, bias, float(s0), float(mult), int(chan_dim))
    return _ChanAffine.apply(_amp(a), w, bias, s0, mult, chan_dim)


_ROWS_ALGO = {"_LokrCore": 0, "_LoconCore": 1, "_LohaCore": 2}  # csrc/torch_ops.cpp ALGO_*


def _rows_conv2d(core, alpha, geom, x, *factors):
    """Conv2d through the row kernels (im2col lowering; the NHWC view for a 1x1 convolution): torch.ops.lycoris_amd.adapter_conv2d
    (C++ dispatch + autograd, traceable) or the Python autograd.Function over the same C ABI calls"""
    if _cpp():
        f2, f3 = (factors[2], factors[3]) if len(factors) == 4 else (None, None)
        return _OPS["adapter_conv2d"](x, factors[0], factors[1], f2, f3, _ROWS_ALGO[core.__name__], float(alpha), list(geom[0]),
                                      list(geom[1]), list(geom[2]), list(geom[3]))
    return _AdapterConv2d.apply(core, alpha, geom, x, *factors)


def _geom(ksize, stride, padding, dilation):
    return (tuple(int(v) for v in ksize), tuple(stride), tuple(padding), tuple(dilation))


def locon_conv2d(x, down, up, alpha, stride, padding, dilation):
    """down:[r, I, kh, kw]  up:[O, r, 1, 1]"""
    r, O = down.shape[0], up.shape[0]
    if not x.is_cuda:
        return _host.locon_conv2d(x, down, up, alpha, stride, padding, dilation)
    x = _amp(x)
    geom = _geom(down.shape[2:], stride, padding, dilation)
    if x.dim() == 4 and not _is_pointwise(geom) and _locon_conv_implicit_ok(x, down, up):
        if _cpp():
            return _OPS["locon_conv2d"](x, down, up, float(alpha), list(geom[1]), list(geom[2]), list(geom[3]))
        return _LoconConv2dImplicit.apply(alpha, geom, x, down, up)
    if (x.dim() == 4 and _is_pointwise(geom) and _cpp() and down.is_contiguous() and up.is_contiguous()
            and not torch.compiler.is_compiling() and x.permute(0, 2, 3, 1).is_contiguous()):
        # 1x1 lora_down on a channels_last tensor: the nn.Linear op on the NHWC pixel rows with the 4-D LEAVES [r, C, 1, 1] / [O, r, 1, 1]
        # (as ops.lokr_conv2d, round 6: a reshaped parameter is a non-leaf view -- no fused accumulation, one factor-gradient launch per layer)
        return _OPS["locon_linear"](x.permute(0, 2, 3, 1), down, up, float(alpha)).permute(0, 3, 1, 2)
    return _rows_conv2d(_LoconCore, alpha, geom, x, down.reshape(r, -1), up.reshape(O, r))


def loha_conv2d(x, w1a, w1b, w2a, w2b, alpha, shape, stride, padding, dilation):
    """w*a:[O, r]  w*b:[r, I*kh*kw];  shape = (O, I, kh, kw)"""
    if not x.is_cuda:
        return _host.loha_conv2d(x, w1a, w1b, w2a, w2b, alpha, shape, stride, padding, dilation)
    return _rows_conv2d(_LohaCore, alpha, _geom(shape[2:], stride, padding, dilation), _amp(x), w1a,
                        w1b.reshape(w1b.shape[0], -1), w2a, w2b.reshape(w2b.shape[0], -1))


def lokr_conv2d(x, w1, w2, alpha, stride, padding, dilation):
    """w1:[a, b]  w2:[c, d, kh, kw].  The channel index u*d + v makes im2col's (channel, kh, kw) column order the
    grouped (u, (v, kh, kw)) order of the Kronecker kernel, so w2 is simply viewed as [c, d*kh*kw]."""
    if not x.is_cuda:
        return _host.lokr_conv2d(x, w1, w2, alpha, stride, padding, dilation)
    x = _amp(x)
    geom = _geom(w2.shape[2:], stride, padding, dilation)
    if x.dim() == 4 and not _is_pointwise(geom) and _lokr_conv_implicit_ok(x, w1, w2):
        if _cpp():
            return _OPS["lokr_conv2d"](x, w1, w2, float(alpha), list(geom[1]), list(geom[2]), list(geom[3]))
        return _LokrConv2dImplicit.apply(alpha, geom, x, w1, w2)
    if (x.dim() == 4 and _is_pointwise(geom) and _cpp() and w2.is_contiguous() and not torch.compiler.is_compiling()
            and x.permute(0, 2, 3, 1).is_contiguous()):  # channels_last (the documented layout); NCHW keeps the lowering's copies and formats
        # a 1x1 convolution IS nn.Linear on the NHWC pixel rows (a free view of a channels_last tensor), and [c, d, 1, 1] is [c, d] in
        # memory: the Linear op takes the 4-D LEAF itself (round 6), so the layer gets everything a Linear layer has -- packed operand
        # planes (kron4 instead of the row kernel), weight gradients accumulated into .grad by the grouped launch.  Reshaping the
        # parameter first (rounds 1 - 5) handed the op a non-leaf view: no planes, no fused accumulation, one dW2 launch per layer.
        return _OPS["lokr_linear"](x.permute(0, 2, 3, 1), w1, w2, float(alpha), None).permute(0, 3, 1, 2)
    return _rows_conv2d(_LokrCore, alpha, geom, x, w1, w2.reshape(w2.shape[0], -1))


def lokr_conv2d_lr(x, w1, w2a, w2b, alpha, ksize, stride, padding, dilation):
    """LoKr on nn.Conv2d with a low-rank second factor (reference modules/lokr.py:131-136): w1:[a, b]  w2a:[c, r]
    w2b:[r, d*kh*kw].  Where the patch kernels take the layer (16-bit activations, leaf fp32 factors, geometry), the operand planes
    are packed from the two factors and the chain rule of the product runs in the grouped weight-gradient launch; everywhere else
    the product is formed (autograd-visible) and handed to lokr_conv2d."""
    ksize, stride, padding, dilation = [int(v) for v in ksize], [int(v) for v in stride], [int(v) for v in padding], [int(v) for v in dilation]
    if (x.is_cuda and _cpp() and not torch.compiler.is_compiling()
            and _DISPATCH["ext"].lokr_conv2d_lr_ok(x, w1, w2a, w2b, ksize, stride, padding, dilation)):
        return _OPS["lokr_conv2d_lr"](x, w1, w2a, w2b, float(alpha), ksize, stride, padding, dilation)
    w2 = (w2a @ w2b).reshape(w2a.shape[0], -1, *ksize)
    return lokr_conv2d(x, w1, w2, alpha, stride, padding, dilation)


# ---------------------------------------------------------------------------------------------------------------
# weight space: merge / diff weight / max-norm / DoRA  (lyc_wspace, lyc_*_wgrad; csrc/wspace.h)
# ---------------------------------------------------------------------------------------------------------------
WS_ALGO = {"locon": 0, "loha": 1, "lokr": 2}
CH_ONE, CH_ROW, CH_COL = 0, 1, 2


def _ws_factors(algo, factors, O, J):
    """fp32 contiguous 2-D views of the factors in the layout lyc_wspace wants, plus (r, a, b, c)."""
    if algo == "locon":  # (down [r, I(,kh,kw)], up [O, r(,1,1)])
        down, up = factors
        r = down.shape[0]
        return [_f32c(down.reshape(r, -1)), _f32c(up.reshape(O, r)), None, None], (r, 0, 0, 0)
    if algo == "loha":   # (w1a [O, r], w1b [r, J], w2a, w2b)
        r = factors[0].shape[1]
        return [_f32c(factors[0]), _f32c(factors[1].reshape(r, -1)), _f32c(factors[2]), _f32c(factors[3].reshape(r, -1))], (r, 0, 0, 0)
    w1, w2 = factors     # lokr: w1 [a, b], w2 [c, d(,kh,kw)]
    a, b = w1.shape
    c = w2.shape[0]
    if a * c != O or J % b:
        raise ValueError(f"LoKr factors {tuple(w1.shape)} x {tuple(w2.shape)} do not tile a [{O}, {J}] weight")
    return [_f32c(w1), _f32c(w2.reshape(c, -1)), None, None], (0, a, b, c)


def _ws_call(algo, fs, dims, O, J, kk, W, w_scale, coef, chan_mode, out, beta, sums, alpha, device):
    r, a, b, c = dims
    N.call("lyc_wspace", WS_ALGO[algo], N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(fs[2]), N.ptr(fs[3]), O, J, r, a, b, c, kk,
           N.ptr(W), N.dtype_code(W.dtype) if W is not None else 0, float(w_scale), N.ptr(coef), chan_mode,
           N.ptr(out), N.dtype_code(out.dtype) if out is not None else 0, float(beta), N.ptr(sums), float(alpha),
           N.stream_ptr(device))


def _wshape(shape):
    O = int(shape[0])
    J = 1
    for s in shape[1:]:
        J *= int(s)
    kk = 1
    for s in shape[2:]:
        kk *= int(s)
    return O, J, kk


@torch.no_grad()
def diff_weight(algo, factors, shape, alpha=1.0, dtype=torch.float32, W=None, coef=None, chan_mode=CH_ONE):
    """alpha * dW (or coef[ch] * (W + alpha * dW) when W / coef are given) as a new [shape] tensor: the dW tile is rebuilt
    on chip, only the result is written (get_diff_weight / get_merged_weight of the reference modules)."""
    O, J, kk = _wshape(shape)
    dev = factors[0].device
    fs, dims = _ws_factors(algo, [f.detach() for f in factors], O, J)
    out = torch.empty(tuple(shape), dtype=dtype, device=dev)
    Wc = None if W is None else W.detach().contiguous()
    _ws_call(algo, fs, dims, O, J, kk, Wc, 1.0, None if coef is None else _f32c(coef).reshape(-1), chan_mode, out, 0.0,
             None, alpha, dev)
    return out


@torch.no_grad()
def merge_into(algo, factors, W, alpha=1.0):
    """W += alpha * dW in place, straight from the factors (merge_to, lycoris/modules/base.py:326-342)."""
    if not W.is_contiguous():
        raise ValueError("merge_into needs a contiguous weight")
    O, J, kk = _wshape(W.shape)
    fs, dims = _ws_factors(algo, [f.detach() for f in factors], O, J)
    _ws_call(algo, fs, dims, O, J, kk, None, 0.0, None, CH_ONE, W, 1.0, None, alpha, W.device)
    return W


@torch.no_grad()
def sq_norm(algo, factors, shape, alpha=1.0):
    """||alpha * dW||_F^2 as a 0-d fp32 tensor, without materialising dW (apply_max_norm, locon.py:273-284)."""
    O, J, kk = _wshape(shape)
    dev = factors[0].device
    fs, dims = _ws_factors(algo, [f.detach() for f in factors], O, J)
    sums = torch.zeros(1, dtype=torch.float32, device=dev)
    _ws_call(algo, fs, dims, O, J, kk, None, 0.0, None, CH_ONE, None, 0.0, sums, alpha, dev)
    return sums[0]


class _WeightNorm2(torch.autograd.Function):
    """norm2[ch] = sum over the channel's slice of (W + alpha * dW)^2 -- DoRA's weight norm (apply_weight_decompose,
    locon.py:239-260) with the dW tile rebuilt on chip: W is read once, nothing [O, J]-sized is written.
    backward: d norm2 / d dW = 2 (W + alpha dW), pushed to the factors through the algorithm's weight-space chain
    rule (lyc_*_wgrad) from a transient fp32 Gw."""

    @staticmethod
    def forward(ctx, algo, W, alpha, chan_mode, *factors):
        N.require_device(W, "base weight")
        O, J, kk = _wshape(W.shape)
        nch = O if chan_mode == CH_ROW else J // kk
        fs, dims = _ws_factors(algo, [f.detach() for f in factors], O, J)
        Wc = W.detach().contiguous()
        sums = torch.zeros(nch, dtype=torch.float32, device=W.device)
        _ws_call(algo, fs, dims, O, J, kk, Wc, 1.0, None, chan_mode, None, 0.0, sums, alpha, W.device)
        ctx.save_for_backward(Wc, *factors)
        ctx.meta = (algo, float(alpha), chan_mode, (O, J, kk))
        return sums

    @staticmethod
    def backward(ctx, gsum):
        algo, alpha, chan_mode, (O, J, kk) = ctx.meta
        Wc, factors = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        need = list(ctx.needs_input_grad[4:])
        if not any(need):
            return (None,) * (4 + len(factors))
        fs, dims = _ws_factors(algo, [f.detach() for f in factors], O, J)
        dev = Wc.device
        coef = (2.0 * gsum).to(torch.float32).contiguous()
        gw = torch.empty((O, J), dtype=torch.float32, device=dev)  # 2 g[ch] (W + alpha dW)
        _ws_call(algo, fs, dims, O, J, kk, Wc, 1.0, coef, chan_mode, gw, 0.0, None, alpha, dev)
        bufs = [torch.zeros(f.shape, dtype=torch.float32, device=dev) for f in fs if f is not None]
        st = N.stream_ptr(dev)
        if algo == "locon":
            N.call("lyc_locon_wgrad", N.ptr(gw), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(bufs[0]), N.ptr(bufs[1]), O, J, dims[0],
                   alpha, st)
        elif algo == "loha":
            N.call("lyc_loha_wgrad", N.ptr(gw), *[N.ptr(f) for f in fs], *[N.ptr(b) for b in bufs], O, J, dims[0], alpha, st)
        else:
            N.call("lyc_lokr_wgrad", N.ptr(gw), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(bufs[0]), N.ptr(bufs[1]), dims[1], dims[2],
                   dims[3], J // dims[2], alpha, st)
        grads = [b.reshape(f.shape).to(f.dtype) if n else None for b, f, n in zip(bufs, factors, need)]
        return (None, None, None, None, *grads)


def weight_norm2(algo, W, factors, alpha=1.0, chan_mode=CH_ROW):
    """differentiable (w.r.t. the factors) squared norms of W + alpha * dW per output row / per input channel"""
    return _WeightNorm2.apply(algo, W, alpha, chan_mode, *factors)


# ---------------------------------------------------------------------------------------------------------------
# Tucker / conv-CP forms: fold the k x k core into the input-side factor (lyc_tucker_core_*; csrc/tucker.h)
# ---------------------------------------------------------------------------------------------------------------
class _TuckerCore(torch.autograd.Function):
    """B[i, q, *k] = sum_j t[i, j, *k] wb[j, q]:  t [r1, r2, kh, kw], wb [r2, Q]  ->  [r1, Q, kh, kw].
    rebuild_tucker(t, wa, wb) (functional/general.py:9-11) == wa^T @ B, so a Tucker adapter is the plain adapter on
    (wa^T, B) and all its activation-path kernels are reused."""

    @staticmethod
    def forward(ctx, t, wb):
        N.require_device(t, "Tucker core")
        tf, wf = _f32c(t), _f32c(wb.reshape(wb.shape[0], -1))
        r1, r2 = tf.shape[0], tf.shape[1]
        kk = 1
        for s in tf.shape[2:]:
            kk *= s
        Q = wf.shape[1]
        if wf.shape[0] != r2:
            raise ValueError(f"tucker_core: core is {tuple(t.shape)}, factor is {tuple(wb.shape)}")
        out = torch.empty((r1, Q, *tf.shape[2:]), dtype=torch.float32, device=t.device)
        N.call("lyc_tucker_core_fwd", N.ptr(tf), N.ptr(wf), N.ptr(out), r1, r2, Q, kk, N.stream_ptr(t.device))
        ctx.save_for_backward(t, wb)
        return out.to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        t, wb = ctx.saved_tensors
        tf, wf = _f32c(t), _f32c(wb.reshape(wb.shape[0], -1))
        r1, r2 = tf.shape[0], tf.shape[1]
        kk = 1
        for s in tf.shape[2:]:
            kk *= s
        Q = wf.shape[1]
        gf = _f32c(g)
        dt = torch.empty_like(tf) if ctx.needs_input_grad[0] else None
        dwb = torch.empty_like(wf) if ctx.needs_input_grad[1] else None
        N.call("lyc_tucker_core_bwd", N.ptr(gf), N.ptr(tf), N.ptr(wf), N.ptr(dt), N.ptr(dwb), r1, r2, Q, kk,
               N.stream_ptr(t.device))
        return (None if dt is None else dt.to(t.dtype)), (None if dwb is None else dwb.reshape(wb.shape).to(wb.dtype))


def tucker_core(t, wb):
    """t [r1, r2, kh, kw] (the Tucker core / lora_mid.weight), wb [r2, Q(,1,1)] -> [r1, Q, kh, kw]"""
    return _TuckerCore.apply(t, wb)
