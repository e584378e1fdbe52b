#!/bin/bash
# round 3, call 11: chunked chain kernel; rank-16 step beside the full-matrix step on the same box; kernel stats of the rank-16 step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -X faulthandler -m pytest tests/test_gpu_custom_ops.py tests/test_gpu_lokr_lowrank.py tests/test_gpu_stress_guard.py -q -k "rows_lowered or lowrank or lokr_lr or chain or conv_op or parked or module_low" --timeout 300 -p no:cacheprovider --maxfail 25 > $O/r03_c11_tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|MISMATCH|OUT-OF" $O/r03_c11_tests.log | cut -c1-300 | head -20
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-base --no-reference --no-roofline --channels-last"
timeout 200 $B > $O/r03_c11_lokr_full.json 2> $O/r03_c11_lokr_full.err; echo "lokr full-matrix rc=$? $(python -c "import json;d=json.loads(open('$O/r03_c11_lokr_full.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
timeout 200 $B --rank 16 > $O/r03_c11_lokr_rank16.json 2> $O/r03_c11_lokr_rank16.err; echo "lokr rank16 rc=$? $(python -c "import json;d=json.loads(open('$O/r03_c11_lokr_rank16.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
timeout 200 $B --rank 16 --layers conv > $O/r03_c11_lokr_rank16_conv.json 2> $O/r03_c11_lokr_rank16_conv.err; echo "lokr rank16 conv-only rc=$? $(python -c "import json;d=json.loads(open('$O/r03_c11_lokr_rank16_conv.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
timeout 200 $B --layers conv > $O/r03_c11_lokr_full_conv.json 2> $O/r03_c11_lokr_full_conv.err; echo "lokr full conv-only rc=$? $(python -c "import json;d=json.loads(open('$O/r03_c11_lokr_full_conv.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/kt_r16 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_r16 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference --no-base --no-roofline --channels-last --rank 16 > $O/r03_c11_prof_rank16.log 2>&1)
f=$(find /tmp/kt_r16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03_c11_rank16_kernel_stats.csv
head -14 $O/r03_c11_rank16_kernel_stats.csv | cut -c1-170
