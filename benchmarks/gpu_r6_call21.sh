#!/bin/bash
# round 6 call 21: LoHa operand-plane cache (one grouped rebuild per optimizer step): tests, LoHa step, kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_loha_planes.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_deferred_wgrad.py tests/test_gpu_modules_golden.py tests/test_gpu_golden_sweep.py tests/test_gpu_stress_guard.py tests/test_gpu_fullsize_oracle.py -m gpu -x -q -k "loha or Loha or stress or sweep or module" > $O/r06_c21_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r06_c21_tests.log
timeout 600 python bench.py --algo loha --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base > $O/r06_c21_bench_loha.json 2> $O/r06_c21_bench_loha.err
python3 -c "
import json;d=json.loads(open('$O/r06_c21_bench_loha.json').read().strip().splitlines()[-1]);r=d.get('roofline') or {};print('loha', d['ms_per_step'], r.get('frac'), r.get('families_ms'))"
tail -2 $O/r06_c21_bench_loha.err | cut -c1-300
