#!/bin/bash
# round 6 call 33: the aborting file order of calls 31 / 32 with the rank-r launches on the rounds 1-5 kernel (A/B: is bneck4 involved?),
# then lcbench's sibling-sum leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
F=$(grep -li "locon" tests/test_gpu_*.py | tr '\n' ' ' | sed 's/tests\/test_gpu_grad_sync.py.*//')
LYC_TEST_LOCON_REG=1 timeout 1500 python -m pytest $F -m gpu -x -q -s > $O/r06_c33_tests_reg.log 2>&1; echo "reg-staged tests rc=$?"
grep -v "^  File\|Extension modules\|^Thread\|no Python frame" $O/r06_c33_tests_reg.log | grep -i "fault\|error\|abort\|passed\|failed" | head -8
cd benchmarks; for n in 3 2; do LCB_SUM=$n timeout 120 ./lcbench attn | grep siblings; LCB_SUM=$n timeout 120 ./lcbench xattn1280 | grep siblings; done
