#!/bin/bash
# where does the --rccl-ws1 step lose its 1.9 ms?  kernel trace of 6 steps: per-step span, busy time, largest gaps, RCCL kernels
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ws1t
E="MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_PORT=29711"
env $E timeout 300 rocprofv3 --kernel-trace -d /tmp/ws1t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-roofline --no-per-algo --rccl-ws1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ws1t/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# the timed steps are the last ones: take the last 40 % of the trace
t0 = rows[0][0]; t1 = rows[-1][1]
cut = t1 - int((t1 - t0) * 0.25)
sel = [r for r in rows if r[0] >= cut]
span = (sel[-1][1] - sel[0][0]) / 1e6
busy = 0; cur_end = sel[0][0]; gaps = []
for s, e, n in sel:
    if s > cur_end:
        gaps.append((s - cur_end, n[:60]))
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
print(f"window {span:.2f} ms, GPU busy (union of kernels) {busy / 1e6:.2f} ms, idle {span - busy / 1e6:.2f} ms, kernels {len(sel)}")
gaps.sort(reverse=True)
print("largest gaps (us, kernel that followed):", [(round(g / 1e3, 1), n) for g, n in gaps[:12]])
rc = [(e - s) / 1e3 for s, e, n in sel if 'ccl' in n.lower()]
print("RCCL kernels in window:", len(rc), "total us", round(sum(rc), 1), "max", round(max(rc), 1) if rc else None)
names = {}
for s, e, n in sel:
    if 'ccl' in n.lower(): names[n[:80]] = names.get(n[:80], 0) + 1
print(names)
PY
