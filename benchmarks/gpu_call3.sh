#!/bin/bash
# round-2 GPU call 3: full parity suite (per-test timeout), host overhead, benches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 700 python -X faulthandler -m pytest tests -m gpu -q --timeout 90 --maxfail 25 -p no:cacheprovider > $O/r02_pytest3.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest3.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/r02_pytest3.log | head -40
timeout 120 python benchmarks/host_overhead.py > $O/r02_host_overhead.log 2>&1; echo "host rc=$?"; tail -4 $O/r02_host_overhead.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r02_bench3_lokr.json 2> $O/r02_bench3_lokr.err; echo "lokr rc=$?"
timeout 200 python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench3_lokr_eager.json 2> $O/r02_bench3_lokr_eager.err; echo "eager rc=$?"
timeout 200 python bench.py --algo ia3 --steps 10 --warmup 3 > $O/r02_bench3_ia3.json 2> $O/r02_bench3_ia3.err; echo "ia3 rc=$?"
timeout 300 python bench.py --algo loha --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench3_loha.json 2> $O/r02_bench3_loha.err; echo "loha rc=$?"
for f in $O/r02_bench3_*.json; do echo "== $f"; head -c 2600 $f; echo; done
for f in $O/r02_bench3_*.err; do echo "== $f"; grep -v "^$" $f | tail -3; done
