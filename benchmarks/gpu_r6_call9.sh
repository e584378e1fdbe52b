#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/prof_loha
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_loha --output-format csv -- python $R/bench.py --algo loha --steps 5 --warmup 1 --no-cpu-baseline --no-reference --no-base --no-per-algo --no-roofline > $O/r06_c9_bench_loha.json 2> $O/r06_c9_bench_loha.err
f=$(find /tmp/prof_loha -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_c9_loha_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/6:9.3f} ms/step {float(r['AverageNs'])/1e3:9.2f} us {float(r['Percentage']):6.2f}%")
PY
