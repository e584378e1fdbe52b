#!/bin/bash
# round 6 call 34: bneck4 with every descriptor-refused offset in the VGPR offset: the new test file, lcbench's sibling-sum leg, the
# aborting file order of calls 31 / 32
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_locon_bneck4.py -m gpu -x -q > $O/r06_c34_tests_bneck4.log 2>&1; echo "bneck4 tests rc=$?"; tail -4 $O/r06_c34_tests_bneck4.log
(cd benchmarks; for n in 3 2; do LCB_SUM=$n timeout 120 ./lcbench attn | grep siblings; done) 2>&1 | tee $O/r06_c34_lcbench_sibling_sum.log
F=$(grep -li "locon" tests/test_gpu_*.py | tr '\n' ' ')
timeout 1800 python -m pytest $F -m gpu -x -q -s > $O/r06_c34_tests_order.log 2>&1; echo "order tests rc=$?"
grep -v "^  File\|Extension modules\|^Thread\|no Python frame" $O/r06_c34_tests_order.log | grep -i "fault\|error\|abort\|passed\|failed" | head -8
