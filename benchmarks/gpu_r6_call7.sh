#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "loha or Loha or stress or fullsize or deferred or functional_api or golden" 2>&1 | tail -15 > $O/r06_c7_loha_tests.log; cat $O/r06_c7_loha_tests.log
timeout 300 python bench.py --algo loha --steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-per-algo > $O/r06_c7_bench_loha.json 2> $O/r06_c7_bench_loha.err; tail -c 1200 $O/r06_c7_bench_loha.json; tail -3 $O/r06_c7_bench_loha.err
