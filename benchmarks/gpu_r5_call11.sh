#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
B="--no-reference --no-base --no-per-algo --no-cpu-baseline --no-roofline --steps 30 --warmup 5"
for v in "plain:" "ws1_inline:--rccl-ws1" "ws1_side:--rccl-ws1 --collectives-on-side-stream" "ws1_captured:--rccl-ws1 --capture-collectives" "ws1_inline_rs:--rccl-ws1 --collective reduce_scatter"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 400 python bench.py $B $flags > gpurun_out/r05_c11_bench_$name.json 2> gpurun_out/r05_c11_bench_$name.err
  echo "$name rc=$? $(tail -1 gpurun_out/r05_c11_bench_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["graph"][40:200])' 2>&1 | cut -c1-300)"
  tail -2 gpurun_out/r05_c11_bench_$name.err | grep -v amdgpu | cut -c1-200
done
timeout 300 python benchmarks/rccl_ws1_check.py --comm rccl > gpurun_out/r05_rccl_ws1_check.log 2>&1; echo "check rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" gpurun_out/r05_rccl_ws1_check.log | tail -6
