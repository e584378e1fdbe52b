#!/usr/bin/env python3
"""A/B (development): the LoKr stage-1 structure of kron3.h (x fragments straight from HBM, fp32 w2 converted to hi/lo per
workgroup) against kron_conv.h (x tile staged in LDS once, pre-packed hi/lo planes streamed by LDS-DMA) on the nn.Linear shapes
of the SDXL step.  The second is reached through the Conv2d entry point with a 1x1 window (the M rows viewed as an H x W image).

    python benchmarks/ab_linear_planes.py            # prints us per launch, forward and backward-dx, both structures
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lycoris_amd import _native as N

DEV = torch.device("cuda:0")
SHAPES = [(1024, 1280, 1280), (1024, 1280, 10240), (1024, 5120, 1280), (4096, 640, 640), (4096, 640, 5120), (4096, 2560, 640),
          (64, 2048, 1280)]


def graph_us(fn, n_inst, reps=5):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e30
        for _ in range(3):
            e0.record(st)
            for _ in range(reps):
                g.replay()
            e1.record(st)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
    torch.cuda.current_stream().wait_stream(st)
    return best * 1e3 / n_inst


def main():
    lib = N.load()
    dtype = torch.bfloat16
    code = N.dtype_code(dtype)
    out = []
    for M, I, O in SHAPES:
        a, c, d = 8, O // 8, I // 8
        ninst = max(4, min(40, int(600e6 / (M * (I + O) * 2))))  # distinct buffers: > the 256 MB infinity cache in total
        H = 1
        while H * H < M:
            H *= 2
        W = M // H if M % H == 0 else None
        if W is None or W % 4 or H % 4:
            H, W = M // 4, 4
        xs = [torch.randn(M, I, device=DEV, dtype=dtype) for _ in range(ninst)]
        gs = [torch.randn(M, O, device=DEV, dtype=dtype) * 0.03 for _ in range(ninst)]
        ys = [torch.empty(M, O, device=DEV, dtype=dtype) for _ in range(ninst)]
        dxs = [torch.empty(M, I, device=DEV, dtype=dtype) for _ in range(ninst)]
        w1 = torch.randn(a, a, device=DEV) * 0.3
        w2 = torch.randn(c, d, device=DEV) * 0.05
        pf = torch.empty(int(lib.lyc_lokr_planes_bytes(c, d, 1, 0)), dtype=torch.uint8, device=DEV)
        pb = torch.empty(int(lib.lyc_lokr_planes_bytes(c, d, 1, 1)), dtype=torch.uint8, device=DEV)
        N.call("lyc_lokr_pack_w2", N.ptr(w2), d, 1, 0, None, 0, 0, None, 0, 0, 0, 0, c, d, 1, N.ptr(pf), N.ptr(pb), code, N.stream_ptr(DEV))
        geo = (1, H, W, a, a, c, d, 1, 1, 1, 1, 0, 0, 1, 1)
        ok = lib.lyc_lokr_conv2d_planes_ok(*geo, code, 0) != 0 and lib.lyc_lokr_conv2d_planes_ok(*geo, code, 1) != 0
        sp = N.stream_ptr

        def rows_fwd():
            for x, y in zip(xs, ys):
                N.call("lyc_lokr_linear_fwd", N.ptr(x), N.ptr(w1), N.ptr(w2), None, N.ptr(y), M, a, a, c, d, 1.0, code, sp(DEV))

        def rows_dx():
            for g, x, dx in zip(gs, xs, dxs):
                N.call("lyc_lokr_linear_bwd", N.ptr(g), N.ptr(x), N.ptr(w1), N.ptr(w2), N.ptr(dx), None, None, None, M, a, a, c, d, 1.0, code, sp(DEV))

        def planes_fwd():
            for x, y in zip(xs, ys):
                N.call("lyc_lokr_conv2d_fwd_planes", N.ptr(x), N.ptr(w1), N.ptr(pf), N.ptr(y), *geo, 1.0, code, sp(DEV))

        def planes_dx():
            for g, x, dx in zip(gs, xs, dxs):
                N.call("lyc_lokr_conv2d_bwd_planes", N.ptr(g), N.ptr(x), N.ptr(w1), None, N.ptr(pb), N.ptr(dx), None, None, None, *geo, 1.0, code, sp(DEV))

        rec = {"M": M, "I": I, "O": O, "instances": ninst, "rows_fwd_us": round(graph_us(rows_fwd, ninst), 2),
               "rows_dx_us": round(graph_us(rows_dx, ninst), 2)}
        if ok:
            ref = ys[0].clone()
            rows_fwd()
            torch.cuda.synchronize()
            ref = ys[0].clone()
            rec.update({"planes_fwd_us": round(graph_us(planes_fwd, ninst), 2), "planes_dx_us": round(graph_us(planes_dx, ninst), 2)})
            torch.cuda.synchronize()
            rec["fwd_rel_diff"] = float((ys[0].float() - ref.float()).norm() / ref.float().norm())
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del xs, gs, ys, dxs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
