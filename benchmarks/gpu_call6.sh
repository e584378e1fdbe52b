#!/bin/bash
# round-2 GPU call 6: fused base+delta, vectorised IA3 kernels, Tucker; benches lokr (base leg) + ia3
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 400 python -X faulthandler -m pytest tests/test_gpu_custom_ops.py tests/test_gpu_linear_ops.py tests/test_gpu_modules_golden.py tests/test_gpu_autocast.py tests/test_gpu_dropout.py -m gpu -q --timeout 90 --maxfail 25 -p no:cacheprovider > $O/r02_pytest6.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest6.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest6.log | head -60
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r02_bench6_lokr.json 2> $O/r02_bench6_lokr.err; echo "lokr rc=$?"
timeout 200 python bench.py --algo ia3 --steps 10 --warmup 3 > $O/r02_bench6_ia3.json 2> $O/r02_bench6_ia3.err; echo "ia3 rc=$?"
for f in $O/r02_bench6_*.json; do echo "== $f"; python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'], json.dumps(d.get('roofline')), json.dumps(d.get('reference_rocm_eager')), json.dumps(d.get('base_plus_adapter')))"; done
for f in $O/r02_bench6_*.err; do echo "== $f"; grep -v "^$" $f | grep -v Warning | tail -3; done
