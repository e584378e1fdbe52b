#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
for f in tests/test_gpu_loha_conv_ops.py tests/test_gpu_custom_ops.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_stress_guard.py tests/test_gpu_modules_golden.py tests/test_gpu_golden_sweep.py tests/test_gpu_conv1d.py tests/test_gpu_functional_api.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -x -q -k "loha or Loha or stress or all_algos or module or rows_lowered or compile" > $O/r06_c10_$n.log 2>&1; echo "$n rc=$?"; grep -a -E "passed|failed|Memory access|^E  " $O/r06_c10_$n.log | tail -6
done
timeout 300 python bench.py --algo loha --steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-per-algo > $O/r06_c10_bench_loha.json 2> $O/r06_c10_bench_loha.err; python3 -c "
import json;d=json.loads(open('$O/r06_c10_bench_loha.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['families_ms'], d['roofline']['frac'])"; tail -3 $O/r06_c10_bench_loha.err
