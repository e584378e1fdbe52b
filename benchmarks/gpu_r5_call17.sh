#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_siblings.py tests/test_gpu_lokr_lowrank.py tests/test_gpu_custom_ops.py -m gpu -x -q > $O/r05_c17_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r05_c17_pytest.log | cut -c1-300
B="--no-reference --no-base --no-per-algo --no-cpu-baseline --no-roofline --steps 30 --warmup 5"
for v in "rank16:--rank 16" "rank16_nosiblings:--rank 16 --no-siblings" "plain:"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 400 python bench.py $B $flags > $O/r05_c17_bench_$name.json 2> $O/r05_c17_bench_$name.err
  echo "$name rc=$? $(tail -1 $O/r05_c17_bench_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>&1 | cut -c1-200)"; tail -2 $O/r05_c17_bench_$name.err | grep -v amdgpu | cut -c1-300
done
