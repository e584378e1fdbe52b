"""Evaluate the numpy oracle on a golden case (shared by CPU and GPU tests)."""
import numpy as np

import oracle


def conv_args_of(meta):
    lk = meta["layer"]
    if lk["kind"] == "linear":
        return None
    return {"stride": lk.get("stride", 1), "padding": lk.get("padding", 0), "dilation": lk.get("dilation", 1)}


def oracle_eval(meta, a, round_dw=None):
    """Returns (delta, grads dict keyed like the golden file: 'dx', 'g.<param name>').
    round_dw (LoHa only): evaluate with the reference's cast of the rebuilt weight (modules/loha.py:310, oracle.loha)."""
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
        mid = a.get("p.lora_mid.weight")
        if mid is not None:  # use_tucker: up(mid(down(x))) = the plain form on the folded input-side factor
            down_raw, down = down, oracle.general.tucker_core(mid, down)
        delta = oracle.locon.forward(x, down, up, eff, ca)
        dx, dd, du = oracle.locon.backward(x, g, down, up, eff, ca)
        out = {"dx": dx, "g.lora_down.weight": dd, "g.lora_up.weight": du}
        if mid is not None:
            out["g.lora_mid.weight"], out["g.lora_down.weight"] = oracle.general.tucker_core_grads(dd, mid, down_raw)
        dscalar = (du * up).sum() / scalar if "p.scalar" in a else None
    elif algo == "loha":
        ws = [a["p.hada_w1_a"], a["p.hada_w1_b"], a["p.hada_w2_a"], a["p.hada_w2_b"]]
        t1, t2 = a.get("p.hada_t1"), a.get("p.hada_t2")
        raw = list(ws)
        if t1 is not None:  # HadaWeightTucker: rebuild_k = w_k_a^T @ fold(t_k, w_k_b)
            r = t1.shape[0]
            ws = [raw[0].T, oracle.general.tucker_core(t1, raw[1]).reshape(r, -1),
                  raw[2].T, oracle.general.tucker_core(t2, raw[3]).reshape(r, -1)]
        delta = oracle.loha.forward(x, *ws, eff, W.shape, ca, round_dw=round_dw)
        dx, g1a, g1b, g2a, g2b = oracle.loha.backward(x, g, *ws, eff, W.shape, ca, round_dw=round_dw)
        if t1 is not None:
            gt1, g1b = oracle.general.tucker_core_grads(g1b, t1, raw[1])
            gt2, g2b = oracle.general.tucker_core_grads(g2b, t2, raw[3])
            g1a, g2a = g1a.T, g2a.T
        out = {"dx": dx, "g.hada_w1_a": g1a, "g.hada_w1_b": g1b, "g.hada_w2_a": g2a, "g.hada_w2_b": g2b}
        if t1 is not None:
            out["g.hada_t1"], out["g.hada_t2"] = gt1, gt2
        dscalar = (g1a * raw[0]).sum() / scalar if "p.scalar" in a else None
    elif algo == "lokr" and "p.lokr_t2" in a:
        t2, w2a, w2b, w1 = a["p.lokr_t2"], a["p.lokr_w2_a"], a["p.lokr_w2_b"], a["p.lokr_w1"]
        r = t2.shape[0]
        fold = oracle.general.tucker_core(t2, w2b)                       # [r, d, kh, kw]
        f2 = (w2a.T @ fold.reshape(r, -1)).reshape(w2a.shape[1], w2b.shape[1], *t2.shape[2:])
        args = dict(w1=w1, w2=f2, scale=eff, kshape=tuple(W.shape[2:]), conv_args=ca)
        delta = oracle.lokr.forward(x, **args)
        gr = oracle.lokr.backward(x, g, **args)
        d2 = gr["w2"].reshape(f2.shape[0], -1)                             # [c, d*kk]
        gt2, gw2b = oracle.general.tucker_core_grads(w2a @ d2, t2, w2b)
        out = {"dx": gr["dx"], "g.lokr_w1": gr["w1"], "g.lokr_w2_a": fold.reshape(r, -1) @ d2.T, "g.lokr_w2_b": gw2b,
               "g.lokr_t2": gt2}
        dscalar = (gr["w1"] * w1).sum() / scalar if "p.scalar" in a else None
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
