#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
B="--no-roofline --no-reference --no-base --no-per-algo --no-cpu-baseline --steps 20 --warmup 3 --host-timing"
for v in "ws1_prio:--rccl-ws1" "ws1_normal:--rccl-ws1 --comm-normal-priority" "ws1_torchstream:--rccl-ws1 --comm-torch-stream" "ws1_rs_torchstream:--rccl-ws1 --comm-torch-stream --collective reduce_scatter"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 300 python bench.py $B $flags > gpurun_out/r05_c4_bench_$name.json 2> gpurun_out/r05_c4_bench_$name.err
  echo "$name rc=$? $(tail -1 gpurun_out/r05_c4_bench_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("host_submit_ms"))' 2>&1 | cut -c1-260)"
done
