#!/bin/bash
# round 6 call 22: LoHa plane cache A/B on one box (--no-planes = per-layer rebuild) + kernel stats of the cached step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_loha_planes.py tests/test_gpu_lokr_planes.py -m gpu -x -q > $O/r06_c22_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_c22_tests.log
for flag in "" "--no-planes" "" "--no-planes"; do
  timeout 600 python bench.py --algo loha --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline $flag > $O/r06_c22_bench_loha$flag.json 2> $O/r06_c22_bench_loha$flag.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c22_bench_loha$flag.json').read().strip().splitlines()[-1]);print('loha $flag', d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_loha
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_loha --output-format csv -- python $R/bench.py --algo loha --steps 5 --warmup 1 --no-cpu-baseline --no-reference --no-base --no-per-algo --no-roofline > $O/r06_c22_prof_loha.json 2> $O/r06_c22_prof_loha.err
f=$(find /tmp/prof_loha -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_c22_loha_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/7:9.3f} ms/step {float(r['AverageNs'])/1e3:9.2f} us")
for r in rows:
    if 'rebuild' in r['Name']: print(r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e6/7, float(r['AverageNs'])/1e3)
PY
