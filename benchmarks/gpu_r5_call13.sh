#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_siblings.py -m gpu -x -q > $O/r05_c13_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r05_c13_pytest.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k1 /tmp/k2
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k1 --output-format csv -- python $R/bench.py --no-reference --no-base --no-per-algo --no-cpu-baseline --no-roofline --steps 10 --warmup 2 > /dev/null 2>&1; echo "prof adapter rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k2 --output-format csv -- python $R/bench.py --dev-base-only > $O/r05_c13_base_only.log 2>&1; echo "prof base rc=$?"; grep "^{" $O/r05_c13_base_only.log | cut -c1-300
f=$(find /tmp/k1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05_c13_adapter_kernel_stats.csv
f=$(find /tmp/k2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r05_c13_base_plus_adapter_kernel_stats.csv
ls -la $O/r05_c13_*csv
