#!/bin/bash
# round 6 call 16: (IA)^3 backward in one pass (lyc_chan_bwd): tests, then the (IA)^3 and mixed steps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_linear_ops.py tests/test_gpu_custom_ops.py tests/test_gpu_modules_golden.py tests/test_gpu_golden_sweep.py -m gpu -x -q -k "chan or ia3 or IA3 or Ia3 or all_algos or sweep" > $O/r06_c16_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r06_c16_tests.log
for algo in ia3 mixed; do
  extra=""; [ $algo = mixed ] && extra="--dtype fp16"
  timeout 600 python bench.py --algo $algo $extra --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base > $O/r06_c16_bench_$algo.json 2> $O/r06_c16_bench_$algo.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c16_bench_$algo.json').read().strip().splitlines()[-1]);r=d.get('roofline') or {};print('$algo', d['ms_per_step'], r.get('frac'), r.get('families_ms'))"; tail -2 $O/r06_c16_bench_$algo.err
done
