#!/bin/bash
# round 6 call 25: kernel stats of the LoKr Conv2d layers alone (bench.py --layers conv)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$PWD/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_conv
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_conv --output-format csv -- python $R/bench.py --algo lokr --layers conv --steps 10 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-per-algo --no-roofline > $O/r06_c25_prof_conv.json 2> $O/r06_c25_prof_conv.err
tail -1 $O/r06_c25_prof_conv.json | cut -c1-300
f=$(find /tmp/prof_conv -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_c25_conv_kernel_stats.csv
t=$(find /tmp/prof_conv -name "*kernel_trace.csv" | head -1); python3 $R/benchmarks/kernel_times.py "$t" lyc > $O/r06_c25_conv_kernel_times.txt
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/13:9.3f} ms/step {float(r['AverageNs'])/1e3:9.2f} us")
PY
