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
idx = [i for i, r in enumerate(rows) if 'oneRankReduce' in r[2] or 'ccl' in r[2].lower()]
print("collective kernels in the trace:", len(idx))
# the last 12 collectives (two timed steps): duration, idle time of the GPU right before it starts and right after it ends
for i in idx[-12:]:
    s, e, n = rows[i]
    prev_end = max(r[1] for r in rows[max(0, i - 40):i]) if i else s
    nxt = next((r[0] for r in rows[i + 1:i + 40] if r[0] >= e), e)
    running = sum(1 for r in rows[max(0, i - 40):i + 40] if r[0] < e and r[1] > s) - 1
    print(f"  {n[:48]:48s} dur {(e - s) / 1e3:7.1f} us | GPU idle before {max(0, s - prev_end) / 1e3:7.1f} us, after {max(0, nxt - e) / 1e3:7.1f} us | kernels overlapping it: {running}")
PY
