#!/bin/bash
# round-2 GPU call 2: weight-space kernels / DoRA / custom-op dispatch parity, host overhead, eager bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $O/r02_pytest2.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest2.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/r02_pytest2.log | head -60
timeout 300 python benchmarks/host_overhead.py > $O/r02_host_overhead.log 2>&1; echo "host rc=$?"; tail -4 $O/r02_host_overhead.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r02_bench2_lokr.json 2> $O/r02_bench2_lokr.err; echo "lokr rc=$?"
timeout 300 python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench2_lokr_eager.json 2> $O/r02_bench2_lokr_eager.err; echo "eager rc=$?"
timeout 300 python bench.py --algo ia3 --steps 10 --warmup 3 > $O/r02_bench2_ia3.json 2> $O/r02_bench2_ia3.err; echo "ia3 rc=$?"
for f in $O/r02_bench2_*.json; do echo "== $f"; head -c 2500 $f; echo; done
for f in $O/r02_bench2_*.err; do echo "== $f"; grep -v "^$" $f | tail -4; done
