#!/bin/bash
# round 6 call 43: gemm16d tile -> XCD mapping: 2-D blocks (gm x gn grid of the 8 L2s over the output) vs the contiguous column-major ranges
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 600 benchmarks/g16bench_1d > $O/r06_c43_g16bench_map1d.log 2>&1; echo "1d rc=$?"
timeout 600 benchmarks/g16bench > $O/r06_c43_g16bench_map2d.log 2>&1; echo "2d rc=$?"
python3 - <<'PY'
import re
def parse(f):
    out={}; cur=None
    for l in open(f):
        m=re.match(r"(\S+)\s+mode (\d)\s+M=(\d+)\s+N=(\d+)\s+K=(\d+)",l)
        if m: cur=(m.group(1),m.group(2)); continue
        m=re.match(r"\s+g16d (\S+ \S+)\s+wgs\s+(\d+).*\|\s+([\d.]+) us.*relerr (\S+)",l)
        if m and cur: out[cur+(m.group(1),)]=(float(m.group(3)),int(m.group(2)),m.group(4))
    return out
a=parse("gpurun_out/r06_c43_g16bench_map1d.log"); b=parse("gpurun_out/r06_c43_g16bench_map2d.log")
for k in a:
    if k in b: print("%-10s mode %s %-12s 1d %7.2f us (%4d wgs)  2d %7.2f us (%4d wgs)  x%.2f  relerr %s" % (k[0],k[1],k[2],a[k][0],a[k][1],b[k][0],b[k][1],a[k][0]/b[k][0],b[k][2]))
PY
