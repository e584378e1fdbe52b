#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
for f in tests/test_gpu_linear_ops.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_stress_guard.py tests/test_gpu_deferred_wgrad.py tests/test_gpu_functional_api.py tests/test_gpu_modules_golden.py tests/test_gpu_golden_sweep.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -x -q -k "loha or Loha or stress or all_algos or module" > $O/r06_c8_$n.log 2>&1; echo "$n rc=$?"; grep -a -E "passed|failed|Memory access|Error|error" $O/r06_c8_$n.log | tail -4
done
