#!/bin/bash
# round 3, call 6: Linear on cached operand planes (kron3 PL), gemm16 with two-deep prefetch, RCCL capture diagnostics, full suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 --maxfail 40 -p no:cacheprovider > $O/r03_c6_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|MISMATCH|OUT-OF" $O/r03_c6_pytest.log | cut -c1-250 | head -30
E="MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
for v in "nopg --no-pg" "skipeager --skip-eager" "eageronly --eager-only"; do set -- $v
  env $E MASTER_PORT=295$((RANDOM % 90 + 10)) timeout 120 python -X faulthandler benchmarks/rccl_ws1_check.py $2 > $O/r03_c6_rccl_$1.log 2>&1; echo "rccl $1 rc=$? $(grep -E 'ok|captured|eager:' $O/r03_c6_rccl_$1.log | tail -2 | tr '\n' ' ' | cut -c1-200)"
done
Q="--steps 10 --warmup 3 --no-cpu-baseline"
timeout 400 python bench.py $Q --channels-last > $O/r03_c6_lokr_cl.json 2> $O/r03_c6_lokr_cl.err; echo "lokr cl rc=$? $(python -c "import json;d=json.load(open('$O/r03_c6_lokr_cl.json'));r=d['roofline'];print(d['ms_per_step'], r['families_ms'], r['achieved'], r['frac'], d.get('reference_rocm_eager'), d.get('base_plus_adapter'))")"
timeout 300 python bench.py $Q --channels-last --no-planes --no-reference --no-base > $O/r03_c6_lokr_cl_noplanes.json 2> $O/r03_c6_lokr_cl_noplanes.err; echo "lokr cl no-planes rc=$? $(python -c "import json;d=json.load(open('$O/r03_c6_lokr_cl_noplanes.json'));r=d['roofline'];print(d['ms_per_step'], r['families_ms'])")"
timeout 300 python bench.py $Q --no-reference --no-base --no-roofline > $O/r03_c6_lokr_nchw.json 2> $O/r03_c6_lokr_nchw.err; echo "lokr nchw rc=$? $(python -c "import json;print(json.load(open('$O/r03_c6_lokr_nchw.json'))['ms_per_step'])")"
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-base --no-reference --algo loha > $O/r03_c6_loha.json 2> $O/r03_c6_loha.err; echo "loha rc=$? $(python -c "import json;d=json.load(open('$O/r03_c6_loha.json'));print(d['ms_per_step'], d.get('roofline',{}).get('families_ms'))")"
export TMPDIR=/tmp
for a in "lokr --channels-last" "loha --algo loha"; do set -- $a; n=$1; shift
  (cd /tmp && rm -rf /tmp/kt_$n && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$n --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference --no-base --no-roofline "$@" > $O/r03_c6_prof_$n.log 2>&1)
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03_c6_${n}_kernel_stats.csv; echo "prof $n: $f"
  grep -v "at::native" $O/r03_c6_${n}_kernel_stats.csv | head -12 | cut -c1-150
done
