#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_siblings.py tests/test_gpu_lokr_group.py tests/test_gpu_grad_sync.py -m gpu -x -q > gpurun_out/r05_c5_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r05_c5_pytest.log | cut -c1-300
B="--no-reference --no-base --no-per-algo --no-cpu-baseline --steps 20 --warmup 3"
for v in "siblings:" "nosiblings:--no-siblings --no-roofline" "ws1:--rccl-ws1 --no-roofline" "ws1_captured:--rccl-ws1 --capture-collectives --no-roofline" "eager:--eager --no-roofline"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 400 python bench.py $B $flags > gpurun_out/r05_c5_bench_$name.json 2> gpurun_out/r05_c5_bench_$name.err
  echo "$name rc=$? $(tail -1 gpurun_out/r05_c5_bench_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print(d["ms_per_step"], r.get("frac"), r.get("avg_launch_us"), r.get("families_ms"))' 2>&1 | cut -c1-400)"
  tail -3 gpurun_out/r05_c5_bench_$name.err | cut -c1-300
done
