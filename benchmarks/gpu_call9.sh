#!/bin/bash
# round-2 GPU call 9: 80x32 tile for the grouped LoKr weight gradients: correctness + sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 200 python -X faulthandler -m pytest tests/test_gpu_deferred_wgrad.py -m gpu -q --timeout 60 --maxfail 10 -p no:cacheprovider > $O/r02_pytest9.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest9.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest9.log | head -40
rm -f $O/r02_wgbench2.log
for v in w1u2 w2u2 w2u1 w0u2 w2t64 w2t256; do
  echo "=== $v" >> $O/r02_wgbench2.log
  timeout 120 ./benchmarks/wgbench_$v >> $O/r02_wgbench2.log 2>&1; echo "rc=$?" >> $O/r02_wgbench2.log
done
cat $O/r02_wgbench2.log
