#!/bin/bash
# round 3, call 1: the parity suite + 200-iteration guard-region stress on the build with the kron3 K<32 fix (never run on a GPU before),
# and the conv / loha baselines this round starts from.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 500 python -X faulthandler -m pytest tests -m gpu -q --timeout 120 --maxfail 25 -p no:cacheprovider > $O/r03_c1_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_c1_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r03_c1_pytest.log | head -20
for cfg in "lokr f16" "lokr bf16" "locon bf16" "locon f16"; do set -- $cfg
  timeout 300 python benchmarks/stress_grouped.py --iters 200 --algo $1 --dtype $2 --seed 11 > $O/r03_c1_stress_$1_$2.log 2>&1; echo "stress $1 $2 rc=$? $(tail -1 $O/r03_c1_stress_$1_$2.log | cut -c1-150)"
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --layers conv --no-reference --no-base > $O/r03_c1_bench_lokr_conv.json 2> $O/r03_c1_bench_lokr_conv.err; echo "conv rc=$?"; cut -c1-400 $O/r03_c1_bench_lokr_conv.json
