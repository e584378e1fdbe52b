#!/bin/bash
# Round 6, first GPU call: the three GPU-untested paths + the five-file fault.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
echo "== (a) sibling sets (autocast default ON), (b) sweep + Conv3d on HIP tensors, n2 check as a 1-rank job =="
timeout 900 python -m pytest tests/test_gpu_siblings.py tests/test_gpu_golden_sweep.py tests/test_gpu_zz_rccl_n2.py -m gpu -q -x 2>&1 | tail -25 > $O/r06_c1_new_tests.log
cat $O/r06_c1_new_tests.log
echo "== (c) five-file order: does this box fault? =="
F="tests/test_gpu_linear_ops.py tests/test_gpu_functional_api.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_modules_golden.py"
timeout 500 python -m pytest $F -m gpu -x -q -s > $O/r06_c1_five_file_plain.log 2>&1; RC=$?; echo "plain rc=$RC"; tail -3 $O/r06_c1_five_file_plain.log
if [ $RC -ne 0 ]; then
  gcc -O1 -g -shared -fPIC benchmarks/abort_bt.c -o benchmarks/abort_bt.so
  echo "== faulting box: synchronous launches + kernel names (AMD_LOG_LEVEL=3 prints every kernel launch; the last one named before the fault is the culprit) =="
  HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 AMD_LOG_LEVEL=3 AMD_LOG_MASK=0x2 MIOPEN_ENABLE_LOGGING_CMD=1 LD_PRELOAD=$PWD/benchmarks/abort_bt.so \
    timeout 900 python -m pytest $F -m gpu -x -q -s -p no:faulthandler > $O/r06_c1_five_file_blocking_full.log 2>&1; echo "blocking rc=$?"
  grep -a -n "ShaderName\|Memory access fault\|abort_bt\|MIOpenDriver" $O/r06_c1_five_file_blocking_full.log | tail -60 > $O/r06_c1_five_file_blocking.log
  tail -c 20000 $O/r06_c1_five_file_blocking_full.log > $O/r06_c1_five_file_blocking_tail.log
  rm -f $O/r06_c1_five_file_blocking_full.log
  tail -40 $O/r06_c1_five_file_blocking.log
fi
echo "== LoHa baseline (per-family ms) and the autocast leg =="
timeout 300 python bench.py --algo loha --steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-per-algo > $O/r06_c1_bench_loha.json 2> $O/r06_c1_bench_loha.err; tail -c 1500 $O/r06_c1_bench_loha.json
timeout 300 python bench.py --algo lokr --autocast --no-roofline --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-base --no-per-algo > $O/r06_c1_bench_autocast.json 2> $O/r06_c1_bench_autocast.err; tail -c 600 $O/r06_c1_bench_autocast.json; tail -5 $O/r06_c1_bench_autocast.err
