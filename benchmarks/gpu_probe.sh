#!/bin/bash
# short GPU probe: parity of the deferred paths + LoHa step A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 300 python -X faulthandler -m pytest tests/test_gpu_deferred_wgrad.py tests/test_gpu_custom_ops.py tests/test_gpu_linear_ops.py -m gpu -q --timeout 60 --maxfail 10 -p no:cacheprovider > $O/r02_pytest_probe.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest_probe.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest_probe.log | head -40
B="--steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-base"
timeout 300 python bench.py --algo loha $B > $O/r02_probe_loha.json 2> $O/r02_probe_loha.err; echo "loha rc=$?"
timeout 300 python bench.py --algo loha $B --no-roofline --no-defer > $O/r02_probe_loha_nodefer.json 2> $O/r02_probe_loha_nodefer.err; echo "loha-nodefer rc=$?"
for f in $O/r02_probe_*.json; do echo "== $f"; python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'], json.dumps(d.get('roofline')))"; done
for f in $O/r02_probe_*.err; do echo "== $f"; grep -v "^$" $f | grep -v Warning | tail -3; done
