#!/bin/bash
# round 5, call 2: the five-file order on a -D_GLIBCXX_ASSERTIONS build (host heap corruption theory) + the ProcessGroup-free RCCL path
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
F="tests/test_gpu_linear_ops.py tests/test_gpu_functional_api.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_modules_golden.py"
echo "== run A: checked build, -s, nothing preloaded =="
timeout 500 python -m pytest $F -m gpu -x -q -s > gpurun_out/r05_abort_runA.log 2>&1
echo "runA rc=$?"; tail -3 gpurun_out/r05_abort_runA.log | cut -c1-300
echo "== run B: MALLOC_CHECK_=3 + backtrace handler =="
gcc -O1 -g -shared -fPIC benchmarks/abort_bt.c -o benchmarks/abort_bt.so
MALLOC_CHECK_=3 MALLOC_PERTURB_=90 LD_PRELOAD=$PWD/benchmarks/abort_bt.so timeout 700 python -m pytest $F -m gpu -x -q -s -p no:faulthandler > gpurun_out/r05_abort_runB.log 2>&1
echo "runB rc=$?"; tail -3 gpurun_out/r05_abort_runB.log | cut -c1-300
grep -n -B8 -A50 "abort_bt\|Assertion\|malloc\|free()\|corrupt" gpurun_out/r05_abort_runA.log gpurun_out/r05_abort_runB.log | head -150
echo "== rccl ws1 check (ProcessGroup-free communicator) =="
timeout 300 python benchmarks/rccl_ws1_check.py --comm rccl > gpurun_out/r05_rccl_ws1_check.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r05_rccl_ws1_check.log
B="--no-roofline --no-reference --no-base --no-per-algo --no-cpu-baseline --steps 30 --warmup 5"
for v in "plain:" "ws1_rccl:--rccl-ws1" "ws1_rccl_captured:--rccl-ws1 --capture-collectives" "ws1_rccl_rs:--rccl-ws1 --collective reduce_scatter" "ws1_c10d:--rccl-ws1 --backend nccl"; do
  name=${v%%:*}; flags=${v#*:}
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py $B $flags > gpurun_out/r05_c2_bench_$name.json 2> gpurun_out/r05_c2_bench_$name.err
  echo "$name rc=$? $(tail -1 gpurun_out/r05_c2_bench_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("graph"))' 2>&1 | cut -c1-260)"
done
