#!/bin/bash
# round 6 call 13: LoHa with rebuild16 + window-major im2col / col2im + split16 factor gradients: LoHa-touching tests, then the step + kernel stats
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
for f in tests/test_gpu_loha_conv_ops.py tests/test_gpu_custom_ops.py tests/test_gpu_deferred_wgrad.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_stress_guard.py tests/test_gpu_modules_golden.py tests/test_gpu_golden_sweep.py tests/test_gpu_conv1d.py tests/test_gpu_functional_api.py tests/test_gpu_linear_ops.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -x -q -k "loha or Loha or stress or all_algos or module or rows_lowered or compile or sweep" > $O/r06_c13_$n.log 2>&1; echo "$n rc=$?"; grep -a -E "passed|failed|Memory access|^E  " $O/r06_c13_$n.log | tail -6
done
cd /tmp
rm -rf /tmp/prof_loha
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_loha --output-format csv -- python $R/bench.py --algo loha --steps 5 --warmup 1 --no-cpu-baseline --no-reference --no-base --no-per-algo --no-roofline > $O/r06_c13_bench_loha.json 2> $O/r06_c13_bench_loha.err
tail -3 $O/r06_c13_bench_loha.err
f=$(find /tmp/prof_loha -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_c13_loha_kernel_stats.csv
python3 - "$f" $O/r06_c13_bench_loha.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/6:9.3f} ms/step {float(r['AverageNs'])/1e3:9.2f} us {float(r['Percentage']):6.2f}%")
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print("ms_per_step", d["ms_per_step"])
PY
