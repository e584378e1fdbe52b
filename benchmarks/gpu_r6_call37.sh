#!/bin/bash
# round 6 call 37: the 8-wave patch kernel: parity (both wave counts), the LoKr conv layers and the headline step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lokr_planes.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_modules_golden.py -m gpu -x -q > $O/r06_c37_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_c37_tests.log
for cfg in "conv:--layers conv" "lokr:"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 600 python bench.py $flags --steps 20 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > $O/r06_c37_bench_$name.json 2> $O/r06_c37_bench_$name.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c37_bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], d['value'])" 2>&1 | tail -1
done
