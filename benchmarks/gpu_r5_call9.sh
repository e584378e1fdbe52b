#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
B="--no-reference --no-base --no-per-algo --no-cpu-baseline --no-roofline --steps 30 --warmup 5"
for v in "plain:" "idle_comm:--dev-idle-comm" "hop_only:--rccl-ws1 --dev-hop-only" "ws1:--rccl-ws1" "hop_only_oneseg:--rccl-ws1 --dev-hop-only --segments 1"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 400 python bench.py $B $flags > gpurun_out/r05_c9_bench_$name.json 2> gpurun_out/r05_c9_bench_$name.err
  echo "$name rc=$? $(tail -1 gpurun_out/r05_c9_bench_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>&1 | cut -c1-300)"
done
