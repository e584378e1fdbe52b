#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_siblings.py tests/test_gpu_linear_ops.py tests/test_gpu_deferred_wgrad.py -m gpu -x -q > $O/r05_c18_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r05_c18_pytest.log | cut -c1-400
B="--no-reference --no-base --no-per-algo --no-cpu-baseline --no-roofline --steps 30 --warmup 5"
for v in "locon:--algo locon" "locon_nosib:--algo locon --no-siblings" "locon_sd15:--algo locon --model sd15" "locon_sd15_nosib:--algo locon --model sd15 --no-siblings" "mixed:--algo mixed --dtype fp16" "mixed_nosib:--algo mixed --dtype fp16 --no-siblings"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 400 python bench.py $B $flags > $O/r05_c18_bench_$name.json 2> $O/r05_c18_bench_$name.err
  echo "$name rc=$? $(tail -1 $O/r05_c18_bench_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>&1 | cut -c1-200)"; tail -2 $O/r05_c18_bench_$name.err | grep -v amdgpu | cut -c1-300
done
