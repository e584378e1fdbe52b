#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_siblings.py tests/test_gpu_lokr_group.py -m gpu -x -q > gpurun_out/r05_c12_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r05_c12_pytest.log | cut -c1-300
B="--no-reference --no-base --no-per-algo --no-cpu-baseline --steps 30 --warmup 5"
timeout 400 python bench.py $B > gpurun_out/r05_c12_bench.json 2> gpurun_out/r05_c12_bench.err
echo "bench rc=$? $(tail -1 gpurun_out/r05_c12_bench.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print(d["ms_per_step"], r.get("frac"), r.get("avg_launch_us"), r.get("families_ms"))' 2>&1 | cut -c1-400)"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_adapter -o adapter -- python bench.py --no-reference --no-base --no-per-algo --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > gpurun_out/r05_c12_prof_adapter.log 2>&1; echo "prof adapter rc=$?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_base -o base -- python bench.py --dev-base-only > gpurun_out/r05_c12_prof_base.log 2>&1; echo "prof base rc=$?"; tail -1 gpurun_out/r05_c12_prof_base.log | cut -c1-300
find gpurun_out/prof_adapter gpurun_out/prof_base -name "*kernel_stats.csv" | head; for f in $(find gpurun_out/prof_adapter gpurun_out/prof_base -name "*kernel_stats.csv"); do cp $f gpurun_out/r05_c12_$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
rm -rf gpurun_out/prof_adapter gpurun_out/prof_base
ls -la gpurun_out/r05_c12_*csv
