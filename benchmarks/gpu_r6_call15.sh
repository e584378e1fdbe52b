#!/bin/bash
# round 6 call 15: lokr_adapted_linear (frozen layer inside the adapter's node): tests, then the base + adapter leg with and without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_adapted_linear.py tests/test_gpu_siblings.py tests/test_gpu_lokr_group.py tests/test_gpu_modules_golden.py tests/test_gpu_autocast.py -m gpu -x -q > $O/r06_c15_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/r06_c15_tests.log
for flag in "" "--no-own-base"; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-roofline $flag > $O/r06_c15_bench_base$flag.json 2> $O/r06_c15_bench_base$flag.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c15_bench_base$flag.json').read().strip().splitlines()[-1]);print('$flag', d['ms_per_step'], d.get('base_plus_adapter'))"; tail -2 $O/r06_c15_bench_base$flag.err
done
