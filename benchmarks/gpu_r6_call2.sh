#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout 900 ./benchmarks/g16bench > $O/r06_c4_g16bench.log 2>&1; echo "rc=$?"
cat $O/r06_c4_g16bench.log
