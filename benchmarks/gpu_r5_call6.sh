#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
B="--no-reference --no-base --no-per-algo --no-cpu-baseline --no-roofline --steps 30 --warmup 5 --host-timing"
for v in "plain:" "segments_only:--force-segments" "ws1:--rccl-ws1" "ws1_rs:--rccl-ws1 --collective reduce_scatter" "ws1_oneseg:--rccl-ws1 --segments 1" "ws1_c10d:--rccl-ws1 --backend nccl" "plain2:"; do
  name=${v%%:*}; flags=${v#*:}
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29579 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py $B $flags > gpurun_out/r05_c6_bench_$name.json 2> gpurun_out/r05_c6_bench_$name.err
  echo "$name rc=$? $(tail -1 gpurun_out/r05_c6_bench_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("host_submit_ms"), d["config"]["graph"][:60])' 2>&1 | cut -c1-300)"
done
