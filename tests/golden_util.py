"""Evaluate the numpy oracle on a golden case (shared by CPU and GPU tests)."""
import numpy as np

import oracle


def conv_args_of(meta):
    lk = meta["layer"]
    if lk["kind"] == "linear":
        return None
    return {"stride": lk.get("stride", 1), "padding": lk.get("padding", 0), "dilation": lk.get("dilation", 1)}


def oracle_eval(meta, a):
    """Returns (delta, grads dict keyed like the golden file: 'dx', 'g.<param name>')."""
    algo = meta["algo"]
    ca = conv_args_of(meta)
    x, g, W = a["x"], a["g"], a["W"]
    scalar = float(a["p.scalar"]) if "p.scalar" in a else 1.0
    eff = meta["scale"] * scalar * meta["multiplier"]
    out = {}
    if meta["mod"].get("weight_decompose"):
        return _oracle_eval_dora(meta, a, scalar, ca)
    if algo == "locon":
        down, up = a["p.lora_down.weight"], a["p.lora_up.weight"]
        delta = oracle.locon.forward(x, down, up, eff, ca)
        dx, dd, du = oracle.locon.backward(x, g, down, up, eff, ca)
        out = {"dx": dx, "g.lora_down.weight": dd, "g.lora_up.weight": du}
        dscalar = (du * up).sum() / scalar if "p.scalar" in a else None
    elif algo == "loha":
        ws = [a["p.hada_w1_a"], a["p.hada_w1_b"], a["p.hada_w2_a"], a["p.hada_w2_b"]]
        delta = oracle.loha.forward(x, *ws, eff, W.shape, ca)
        dx, g1a, g1b, g2a, g2b = oracle.loha.backward(x, g, *ws, eff, W.shape, ca)
        out = {"dx": dx, "g.hada_w1_a": g1a, "g.hada_w1_b": g1b, "g.hada_w2_a": g2a, "g.hada_w2_b": g2b}
        dscalar = (g1a * ws[0]).sum() / scalar if "p.scalar" in a else None
    elif algo == "lokr":
        kw = {k: a.get("p.lokr_" + k) for k in ("w1", "w1_a", "w1_b", "w2", "w2_a", "w2_b")}
        args = dict(w1=kw["w1"], w1a=kw["w1_a"], w1b=kw["w1_b"], w2=kw["w2"], w2a=kw["w2_a"], w2b=kw["w2_b"],
                    scale=eff, kshape=tuple(W.shape[2:]), conv_args=ca)
        delta = oracle.lokr.forward(x, **args)
        gr = oracle.lokr.backward(x, g, **args)
        out = {"dx": gr.pop("dx")}
        for k, v in gr.items():
            out["g.lokr_" + k.replace("w1a", "w1_a").replace("w1b", "w1_b").replace("w2a", "w2_a").replace("w2b", "w2_b")] = v
        k1 = "w1" if kw["w1"] is not None else "w1_a"
        dscalar = (out["g.lokr_" + k1] * kw[k1]).sum() / scalar if "p.scalar" in a else None
    elif algo == "ia3":
        on_input = bool(meta["mod"].get("train_on_input", False))
        w = a["p.weight"]
        delta = oracle.ia3.forward(x, W, w, meta["multiplier"], on_input, ca)
        dx, dw = oracle.ia3.backward(x, g, W, w, meta["multiplier"], on_input, ca)
        out = {"dx": dx, "g.weight": dw}
        dscalar = None
    else:
        raise KeyError(algo)
    if dscalar is not None:
        out["g.scalar"] = np.asarray(dscalar)
    return delta, out


def _oracle_eval_dora(meta, a, scalar, ca):
    """weight_decompose=True cases: dW with scale * scalar (no multiplier), then oracle.dora."""
    algo = meta["algo"]
    x, g, W = a["x"], a["g"], a["W"]
    sc = meta["scale"] * scalar
    on_out = bool(meta["mod"].get("wd_on_out", True))
    mult = meta["multiplier"]
    if algo == "locon":
        down, up = a["p.lora_down.weight"], a["p.lora_up.weight"]
        dW = oracle.locon.diff_weight(down, up, sc).reshape(W.shape)
    elif algo == "loha":
        ws = [a["p.hada_w1_a"], a["p.hada_w1_b"], a["p.hada_w2_a"], a["p.hada_w2_b"]]
        dW = oracle.loha.diff_weight(*ws, sc, W.shape)
    else:
        kw = {k: a.get("p.lokr_" + k) for k in ("w1", "w1_a", "w1_b", "w2", "w2_a", "w2_b")}
        fa = dict(w1=kw["w1"], w1a=kw["w1_a"], w1b=kw["w1_b"], w2=kw["w2"], w2a=kw["w2_a"], w2b=kw["w2_b"])
        dW = oracle.lokr.diff_weight(**fa, scale=sc, kshape=tuple(W.shape[2:])).reshape(W.shape)
    ds = a["p.dora_scale"]
    delta = oracle.dora.forward(x, W, dW, ds, mult, on_out, ca)
    dx, dV, d_ds = oracle.dora.backward(x, g, W, dW, ds, mult, on_out, ca)
    out = {"dx": dx, "g.dora_scale": d_ds}
    if algo == "locon":
        dd, du = oracle.locon.factor_grads(dV, down, up, sc)
        out.update({"g.lora_down.weight": dd, "g.lora_up.weight": du})
        first = (du, up)
    elif algo == "loha":
        gs = oracle.loha.factor_grads(dV, *ws, sc)
        out.update(dict(zip(["g.hada_w1_a", "g.hada_w1_b", "g.hada_w2_a", "g.hada_w2_b"], gs)))
        first = (gs[0], ws[0])
    else:
        gr = oracle.lokr.factor_grads(dV, **fa, scale=sc, kshape=tuple(W.shape[2:]))
        for k, v in gr.items():
            out["g.lokr_" + k.replace("w1a", "w1_a").replace("w1b", "w1_b").replace("w2a", "w2_a").replace("w2b", "w2_b")] = v
        k1 = "w1" if kw["w1"] is not None else "w1_a"
        first = (out["g.lokr_" + k1], kw[k1])
    if "p.scalar" in a:
        out["g.scalar"] = np.asarray((first[0] * first[1]).sum() / scalar)
    return delta, out
