#!/bin/bash
# round 3, call 12: the whole -m gpu suite on the current build; eager host profile after the batched notification
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --maxfail 30 > $O/r03_c12_pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/r03_c12_pytest_gpu.log | cut -c1-300 | head -40
timeout 200 python benchmarks/eager_profile.py 2>&1 | grep -E "^layers" 
