#!/bin/bash
# round-2 GPU call 7: grouped LoKr weight gradients (A/B), LoHa staging fix, full parity suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 200 python -X faulthandler -m pytest tests/test_gpu_deferred_wgrad.py tests/test_gpu_custom_ops.py -m gpu -q --timeout 60 --maxfail 10 -p no:cacheprovider > $O/r02_pytest7a.log 2>&1; echo "pytest-a rc=$?" | tee -a $O/r02_pytest7a.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest7a.log | head -40
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-base > $O/r02_bench7_lokr.json 2> $O/r02_bench7_lokr.err; echo "lokr rc=$?"
timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-base --no-roofline --no-defer > $O/r02_bench7_lokr_nodefer.json 2> $O/r02_bench7_lokr_nodefer.err; echo "lokr-nodefer rc=$?"
timeout 150 python bench.py --steps 10 --warmup 3 --eager > $O/r02_bench7_lokr_eager.json 2> $O/r02_bench7_lokr_eager.err; echo "lokr-eager rc=$?"
timeout 200 python bench.py --algo loha --steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-base > $O/r02_bench7_loha.json 2> $O/r02_bench7_loha.err; echo "loha rc=$?"
for f in $O/r02_bench7_*.json; do echo "== $f"; python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'], json.dumps(d.get('roofline')))"; done
for f in $O/r02_bench7_*.err; do echo "== $f"; grep -v "^$" $f | grep -v Warning | tail -3; done
timeout 500 python -X faulthandler -m pytest tests -m gpu -q --timeout 90 --maxfail 25 -p no:cacheprovider --deselect tests/test_gpu_deferred_wgrad.py --deselect tests/test_gpu_custom_ops.py > $O/r02_pytest7b.log 2>&1; echo "pytest-b rc=$?" | tee -a $O/r02_pytest7b.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest7b.log | head -40
