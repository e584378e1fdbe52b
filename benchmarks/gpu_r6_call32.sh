#!/bin/bash
# round 6 call 32: the file order of call 31 again with -s (the abort's message), then the files behind the aborting one
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
F=$(grep -li "locon" tests/test_gpu_*.py | tr '\n' ' ')
timeout 1500 python -m pytest $F -m gpu -x -q -s > $O/r06_c32_tests_s.log 2>&1; echo "tests rc=$?"
grep -v "^  File\|Extension modules\|^Thread\|no Python frame" $O/r06_c32_tests_s.log | grep -i "fault\|error\|abort\|passed\|failed\|HSA\|gpu" | head -20
R="tests/test_gpu_grad_sync.py tests/test_gpu_linear_ops.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_modules_golden.py tests/test_gpu_parity_round4.py tests/test_gpu_siblings.py tests/test_gpu_stress_guard.py tests/test_gpu_wspace.py"
timeout 1500 python -m pytest $R -m gpu -x -q > $O/r06_c32_tests_rest.log 2>&1; echo "rest rc=$?"; tail -3 $O/r06_c32_tests_rest.log
