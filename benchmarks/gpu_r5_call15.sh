#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/r05_c15_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/r05_c15_pytest_gpu.log | cut -c1-300
