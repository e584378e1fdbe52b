#!/bin/bash
# round 6 call 14: kernel stats of the LoKr headline step alone (no measurement legs), and of the LoCon / (IA)^3 steps
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
for algo in lokr locon ia3; do
rm -rf /tmp/prof_$algo
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$algo --output-format csv -- python $R/bench.py --algo $algo --steps 10 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-per-algo --no-roofline > $O/r06_c14_bench_$algo.json 2> $O/r06_c14_bench_$algo.err
tail -2 $O/r06_c14_bench_$algo.err
f=$(find /tmp/prof_$algo -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_c14_${algo}_kernel_stats.csv
python3 - "$f" $O/r06_c14_bench_$algo.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print("ms_per_step", d["ms_per_step"])
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / 13
print("kernel time per pass (13 passes assumed)", round(tot, 3))
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/13:9.3f} ms/step {float(r['AverageNs'])/1e3:9.2f} us {float(r['Percentage']):6.2f}%")
PY
done
