#!/bin/bash
# round 3, call 7: gemm16 with swizzled LDS images; segmented-capture diagnostics; low-rank leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 500 python -X faulthandler -m pytest tests/test_gpu_loha_conv_ops.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_deferred_wgrad.py tests/test_gpu_stress_guard.py -q -k "loha" --timeout 300 -p no:cacheprovider --maxfail 15 > $O/r03_c7_loha_tests.log 2>&1; echo "loha tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|MISMATCH|OUT-OF" $O/r03_c7_loha_tests.log | cut -c1-250 | head
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-base --algo loha > $O/r03_c7_loha.json 2> $O/r03_c7_loha.err; echo "loha rc=$? $(python -c "import json;d=json.load(open('$O/r03_c7_loha.json'));print(d['ms_per_step'], d.get('roofline',{}).get('families_ms'), d.get('reference_rocm_eager'))")"
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-base --no-reference --no-roofline --channels-last --force-segments --segments 6 > $O/r03_c7_lokr_segments.json 2> $O/r03_c7_lokr_segments.err; echo "lokr forced segments rc=$? $(python -c "import json;d=json.load(open('$O/r03_c7_lokr_segments.json'));print(d['ms_per_step'], d['config']['graph'])")"; tail -3 $O/r03_c7_lokr_segments.err | cut -c1-200
E="MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
for v in "nopg_noshared --no-pg --no-shared" "nopg --no-pg" "full"; do set -- $v; n=$1; shift
  env $E MASTER_PORT=295$((RANDOM % 90 + 10)) timeout 120 python -X faulthandler benchmarks/rccl_ws1_check.py "$@" > $O/r03_c7_rccl_$n.log 2>&1; echo "rccl $n rc=$? $(grep -E 'ok|captured|eager:' $O/r03_c7_rccl_$n.log | tail -3 | tr '\n' ' ' | cut -c1-260)"
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-base --channels-last --rank 16 > $O/r03_c7_lokr_rank16.json 2> $O/r03_c7_lokr_rank16.err; echo "lokr rank16 rc=$? $(python -c "import json;d=json.load(open('$O/r03_c7_lokr_rank16.json'));print(d['ms_per_step'], d['config']['adapter_params'], d.get('reference_rocm_eager'))")"
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/kt_loha && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_loha --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference --no-base --no-roofline --algo loha > $O/r03_c7_prof_loha.log 2>&1)
f=$(find /tmp/kt_loha -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03_c7_loha_kernel_stats.csv
grep -v "at::native" $O/r03_c7_loha_kernel_stats.csv | head -10 | cut -c1-150
