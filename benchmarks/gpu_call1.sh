#!/bin/bash
# round-2 GPU call 1: parity suite + first bench lines of every algorithm with the rotated-buffer workload + PMC traffic
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r02_pytest1.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest1.log
tail -5 $O/r02_pytest1.log
timeout 400 python bench.py --steps 10 --warmup 3 > $O/r02_bench_lokr.json 2> $O/r02_bench_lokr.err; echo "lokr rc=$?"
timeout 300 python bench.py --steps 10 --warmup 3 --shared-inputs --no-cpu-baseline --no-reference --no-base > $O/r02_bench_lokr_shared.json 2> $O/r02_bench_lokr_shared.err; echo "lokr shared rc=$?"
timeout 300 python bench.py --algo locon --steps 10 --warmup 3 --no-cpu-baseline > $O/r02_bench_locon.json 2> $O/r02_bench_locon.err; echo "locon rc=$?"
timeout 400 python bench.py --algo loha --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_loha.json 2> $O/r02_bench_loha.err; echo "loha rc=$?"
timeout 300 python bench.py --algo ia3 --steps 10 --warmup 3 > $O/r02_bench_ia3.json 2> $O/r02_bench_ia3.err; echo "ia3 rc=$?"
timeout 300 python bench.py --preset mixed --dtype fp16 --steps 10 --warmup 3 > $O/r02_bench_mixed.json 2> $O/r02_bench_mixed.err; echo "mixed rc=$?"
timeout 300 python bench.py --model sd15 --algo locon --steps 10 --warmup 3 --no-cpu-baseline > $O/r02_bench_sd15_locon.json 2> $O/r02_bench_sd15_locon.err; echo "sd15 rc=$?"
timeout 300 python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_lokr_eager.json 2> $O/r02_bench_lokr_eager.err; echo "eager rc=$?"
ALGOS="lokr locon" timeout 900 bash benchmarks/pmc_traffic.sh > $O/r02_pmc.log 2>&1; echo "pmc rc=$?"
for f in $O/r02_bench_*.json; do echo "== $f"; head -c 1500 $f; echo; done
tail -3 $O/r02_bench_*.err | tail -40
