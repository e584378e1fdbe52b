#!/bin/bash
# round-2 GPU call 8: plan / prefetch-depth sweep of the grouped LoKr weight-gradient launch (benchmarks/wgbench.cpp)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
for v in base t64 t256 big u2 u1; do
  echo "=== $v" >> $O/r02_wgbench.log
  timeout 120 ./benchmarks/wgbench_$v >> $O/r02_wgbench.log 2>&1; echo "rc=$?" >> $O/r02_wgbench.log
done
cat $O/r02_wgbench.log
