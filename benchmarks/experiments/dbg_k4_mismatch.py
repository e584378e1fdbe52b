import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lycoris_amd import _native as N
DEV = torch.device("cuda:0")
lib = N.load()
def run(M, a, c, d, dtype, alpha, reps=50, seed=0):
    gen = torch.Generator().manual_seed(seed)
    code = N.dtype_code(dtype)
    x = (torch.randn(M, a * d, generator=gen) * 0.5).to(dtype).to(DEV)
    w1 = (torch.randn(a, a, generator=gen) * 0.3).to(DEV)
    w2 = (torch.randn(c, d, generator=gen) * 0.1).to(DEV)
    nf, nb = int(lib.lyc_lokr_planes_bytes(c, d, 1, 0)), int(lib.lyc_lokr_planes_bytes(c, d, 1, 1))
    pf = torch.empty(nf, dtype=torch.uint8, device=DEV); pb = torch.empty(nb, dtype=torch.uint8, device=DEV)
    N.call("lyc_lokr_pack_w2", N.ptr(w2), d, 1, 0, None, 0, 0, None, 0, 0, 0, 0, c, d, 1, N.ptr(pf), N.ptr(pb), code, N.stream_ptr(DEV))
    y = torch.empty(M, a * c, dtype=dtype, device=DEV)
    N.call("lyc_lokr_linear_fwd", N.ptr(x), N.ptr(w1), N.ptr(w2), None, N.ptr(y), M, a, a, c, d, alpha, code, N.stream_ptr(DEV))
    torch.cuda.synchronize()
    bad = 0
    first = None
    for r in range(reps):
        yp = torch.full((M, a * c), float("nan"), dtype=dtype, device=DEV)
        N.call("lyc_lokr_linear_fwd_planes", N.ptr(x), N.ptr(w1), N.ptr(pf), None, N.ptr(yp), M, a, a, c, d, alpha, code, N.stream_ptr(DEV))
        torch.cuda.synchronize()
        ne = (yp != y) | torch.isnan(yp)
        if bool(ne.any()):
            bad += 1
            if first is None:
                idx = ne.nonzero()
                first = (int(ne.sum()), idx[:6].tolist(), [(float(yp[i, j]), float(y[i, j])) for i, j in idx[:6].tolist()])
    print(f"M={M} a={a} c={c} d={d} {dtype} alpha={alpha}: {bad}/{reps} runs differ; first: {first}", flush=True)
for dt in (torch.float16, torch.bfloat16):
    run(77, 8, 88, 80, dt, 0.7)
    run(77, 8, 88, 80, dt, 0.5)
    run(77, 8, 96, 80, dt, 0.7)
    run(77, 8, 88, 96, dt, 0.7)
    run(1024, 8, 160, 160, dt, 0.7)
    run(1024, 8, 160, 160, dt, 0.5)
    run(333, 4, 40, 72, dt, 0.7)
