#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/r05_c14_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/r05_c14_pytest_gpu.log | cut -c1-300
echo "== python bench.py --gpus 2 (no launcher around it): two ranks on the one GPU through gloo =="
timeout 400 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline > $O/r05_c14_n2_gloo.json 2> $O/r05_c14_n2_gloo.err; echo "n2 rc=$?"
tail -1 $O/r05_c14_n2_gloo.json | cut -c1-600; tail -3 $O/r05_c14_n2_gloo.err | cut -c1-300
