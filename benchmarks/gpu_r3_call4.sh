#!/bin/bash
# round 3, call 4: patch staged by LDS-DMA; grouped conv dW2; RCCL ws1 with traces; full suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 400 python -X faulthandler -m pytest tests/test_gpu_lokr_planes.py tests/test_gpu_grad_sync.py "tests/test_gpu_stress_guard.py" -q --timeout 250 -p no:cacheprovider --maxfail 10 > $O/r03_c4_new.log 2>&1; echo "new tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|MISMATCH|OUT-OF" $O/r03_c4_new.log | cut -c1-300 | head -20
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 200 python -X faulthandler benchmarks/rccl_ws1_check.py > $O/r03_c4_rccl.log 2>&1; echo "rccl rc=$?"; grep -v "Warning\|warn" $O/r03_c4_rccl.log | tail -25 | cut -c1-200
Q="--steps 10 --warmup 3 --no-cpu-baseline --layers conv --no-reference --no-base --no-roofline"
timeout 300 python bench.py $Q > $O/r03_c4_conv_nchw.json 2> $O/r03_c4_conv_nchw.err; echo "conv nchw rc=$? $(python -c "import json;print(json.load(open('$O/r03_c4_conv_nchw.json'))['ms_per_step'])")"
timeout 300 python bench.py $Q --channels-last > $O/r03_c4_conv_cl.json 2> $O/r03_c4_conv_cl.err; echo "conv cl rc=$? $(python -c "import json;print(json.load(open('$O/r03_c4_conv_cl.json'))['ms_per_step'])")"
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/kt_conv && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_conv --output-format csv -- python $R/bench.py --steps 3 --warmup 1 $Q --channels-last > $O/r03_c4_prof_conv.log 2>&1)
f=$(find /tmp/kt_conv -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03_c4_conv_cl_kernel_stats.csv; echo "prof: $f"
grep -v "at::native" $O/r03_c4_conv_cl_kernel_stats.csv | head -22 | cut -c1-160
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 --maxfail 40 -p no:cacheprovider --deselect tests/test_gpu_lokr_planes.py --deselect tests/test_gpu_grad_sync.py --deselect tests/test_gpu_stress_guard.py > $O/r03_c4_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $O/r03_c4_pytest.log | cut -c1-250 | head -30
