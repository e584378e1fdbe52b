#!/bin/bash
# round 3, call 3: patch kernel with MI = 2 / 4 / 8 row tiles, backward on planes; A/B of the stage-1 structures on the Linear shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 400 python -X faulthandler -m pytest tests/test_gpu_lokr_planes.py tests/test_gpu_grad_sync.py "tests/test_gpu_stress_guard.py" -q --timeout 250 -p no:cacheprovider --maxfail 10 > $O/r03_c3_new.log 2>&1; echo "new tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|MISMATCH|OUT-OF" $O/r03_c3_new.log | cut -c1-300 | head -20
timeout 300 python -X faulthandler -m pytest tests/test_gpu_fullsize_oracle.py -q -k "conv" --timeout 250 -p no:cacheprovider --maxfail 10 > $O/r03_c3_fullconv.log 2>&1; echo "fullsize conv rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/r03_c3_fullconv.log | cut -c1-300 | head -20
Q="--steps 10 --warmup 3 --no-cpu-baseline --layers conv --no-reference --no-base --no-roofline"
timeout 300 python bench.py $Q > $O/r03_c3_conv_nchw.json 2> $O/r03_c3_conv_nchw.err; echo "conv nchw rc=$? $(python -c "import json;print(json.load(open('$O/r03_c3_conv_nchw.json'))['ms_per_step'])")"
timeout 300 python bench.py $Q --channels-last > $O/r03_c3_conv_cl.json 2> $O/r03_c3_conv_cl.err; echo "conv cl rc=$? $(python -c "import json;print(json.load(open('$O/r03_c3_conv_cl.json'))['ms_per_step'])")"
timeout 300 python benchmarks/ab_linear_planes.py > $O/r03_c3_ab_linear.log 2>&1; echo "ab rc=$?"; cat $O/r03_c3_ab_linear.log | cut -c1-300
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/kt_conv && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_conv --output-format csv -- python $R/bench.py --steps 3 --warmup 1 $Q --channels-last > $O/r03_c3_prof_conv.log 2>&1)
f=$(find /tmp/kt_conv -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03_c3_conv_cl_kernel_stats.csv; echo "prof: $f"
head -16 $O/r03_c3_conv_cl_kernel_stats.csv | cut -c1-160
