#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_siblings.py tests/test_gpu_lokr_group.py tests/test_gpu_deferred_wgrad.py -m gpu -x -q > $O/r05_c16_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r05_c16_pytest.log | cut -c1-300
B="--no-reference --no-per-algo --no-cpu-baseline --steps 30 --warmup 5"
timeout 400 python bench.py $B > $O/r05_c16_bench.json 2> $O/r05_c16_bench.err
echo "bench rc=$? $(tail -1 $O/r05_c16_bench.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print(d["ms_per_step"], r.get("frac"), r.get("avg_launch_us"), r.get("families_ms"), d.get("base_plus_adapter"))' 2>&1 | cut -c1-700)"
