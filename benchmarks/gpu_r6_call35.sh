#!/bin/bash
# round 6 call 35: the whole GPU suite on the library with bneck4 (single, group, fused sibling sum), then the LoCon-bearing bench legs
# and the kernel stats of the LoCon step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$PWD/gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r06_c35_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r06_c35_pytest_gpu.log | cut -c1-300
for cfg in "locon:--algo locon" "locon_sd15:--algo locon --model sd15" "mixed:--algo mixed --dtype fp16"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 600 python bench.py $flags --steps 20 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > $O/r06_c35_bench_$name.json 2> $O/r06_c35_bench_$name.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c35_bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], d['value'])" 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_locon
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_locon --output-format csv -- python $R/bench.py --algo locon --steps 10 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-per-algo --no-roofline > $O/r06_c35_prof_locon.json 2> $O/r06_c35_prof_locon.err
f=$(find /tmp/prof_locon -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_c35_locon_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:110]:110s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/13:9.3f} ms/step {float(r['AverageNs'])/1e3:9.2f} us")
PY
