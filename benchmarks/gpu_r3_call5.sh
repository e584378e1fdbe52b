#!/bin/bash
# round 3, call 5: patch dW2 conv kernel + gemm16 (LoHa without rocBLAS): parity, stress, benches, kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -X faulthandler -m pytest tests/test_gpu_lokr_planes.py tests/test_gpu_grad_sync.py tests/test_gpu_stress_guard.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_deferred_wgrad.py -q --timeout 300 -p no:cacheprovider --maxfail 15 > $O/r03_c5_new.log 2>&1; echo "new tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|MISMATCH|OUT-OF" $O/r03_c5_new.log | cut -c1-300 | head -30
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-base --no-roofline"
timeout 300 python bench.py $Q --layers conv --channels-last > $O/r03_c5_conv_cl.json 2> $O/r03_c5_conv_cl.err; echo "conv cl rc=$? $(python -c "import json;print(json.load(open('$O/r03_c5_conv_cl.json'))['ms_per_step'])")"
LYC_CONV_DW2_ROWS=1 timeout 300 python bench.py $Q --layers conv --channels-last > $O/r03_c5_conv_cl_rowsdw2.json 2> $O/r03_c5_conv_cl_rowsdw2.err; echo "conv cl (row-gather dW2) rc=$? $(python -c "import json;print(json.load(open('$O/r03_c5_conv_cl_rowsdw2.json'))['ms_per_step'])")"
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-base --algo loha > $O/r03_c5_loha.json 2> $O/r03_c5_loha.err; echo "loha rc=$? $(python -c "import json;d=json.load(open('$O/r03_c5_loha.json'));print(d['ms_per_step'], d.get('roofline',{}).get('families_ms'), d.get('reference_rocm_eager'))")"
timeout 300 python bench.py $Q --channels-last > $O/r03_c5_lokr_cl.json 2> $O/r03_c5_lokr_cl.err; echo "lokr cl rc=$? $(python -c "import json;print(json.load(open('$O/r03_c5_lokr_cl.json'))['ms_per_step'])")"
export TMPDIR=/tmp
for a in "conv --layers conv --channels-last" "loha --algo loha"; do set -- $a; n=$1; shift
  (cd /tmp && rm -rf /tmp/kt_$n && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$n --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference --no-base --no-roofline "$@" > $O/r03_c5_prof_$n.log 2>&1)
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03_c5_${n}_kernel_stats.csv; echo "prof $n: $f"
  grep -v "at::native" $O/r03_c5_${n}_kernel_stats.csv | head -14 | cut -c1-150
done
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 --maxfail 40 -p no:cacheprovider --deselect tests/test_gpu_lokr_planes.py --deselect tests/test_gpu_grad_sync.py --deselect tests/test_gpu_stress_guard.py --deselect tests/test_gpu_loha_conv_ops.py --deselect tests/test_gpu_deferred_wgrad.py > $O/r03_c5_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $O/r03_c5_pytest.log | cut -c1-250 | head -30
