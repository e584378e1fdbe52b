#!/bin/bash
# round 3, call 2: first run of the LDS-patch Conv2d kernels on packed planes: new parity tests, the whole suite, conv-only bench legs, kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 300 python -X faulthandler -m pytest tests/test_gpu_lokr_planes.py tests/test_gpu_grad_sync.py -q --timeout 200 -p no:cacheprovider -x > $O/r03_c2_new.log 2>&1; echo "new tests rc=$?"; tail -15 $O/r03_c2_new.log | cut -c1-300
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 --maxfail 40 -p no:cacheprovider --deselect tests/test_gpu_lokr_planes.py --deselect tests/test_gpu_grad_sync.py > $O/r03_c2_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $O/r03_c2_pytest.log | cut -c1-250 | head -50
Q="--steps 10 --warmup 3 --no-cpu-baseline --layers conv --no-reference --no-base --no-roofline"
timeout 300 python bench.py $Q > $O/r03_c2_conv_nchw.json 2> $O/r03_c2_conv_nchw.err; echo "conv nchw rc=$? $(python -c "import json;print(json.load(open('$O/r03_c2_conv_nchw.json'))['ms_per_step'])")"
timeout 300 python bench.py $Q --channels-last > $O/r03_c2_conv_cl.json 2> $O/r03_c2_conv_cl.err; echo "conv cl rc=$? $(python -c "import json;print(json.load(open('$O/r03_c2_conv_cl.json'))['ms_per_step'])")"
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/kt_conv && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_conv --output-format csv -- python $R/bench.py --steps 3 --warmup 1 $Q --channels-last > $O/r03_c2_prof_conv.log 2>&1)
f=$(find /tmp/kt_conv -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03_c2_conv_cl_kernel_stats.csv; echo "prof: $f"
head -25 $O/r03_c2_conv_cl_kernel_stats.csv | cut -c1-200
