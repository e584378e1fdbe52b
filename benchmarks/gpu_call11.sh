#!/bin/bash
# round-2 GPU call 11: grouped LoCon factor gradients (parity + A/B), LoKr with the 80x32 grouped tile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 300 python -X faulthandler -m pytest tests/test_gpu_deferred_wgrad.py tests/test_gpu_custom_ops.py tests/test_gpu_linear_ops.py -m gpu -q --timeout 60 --maxfail 10 -p no:cacheprovider > $O/r02_pytest11.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest11.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest11.log | head -40
B="--steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-base"
timeout 200 python bench.py $B > $O/r02_bench11_lokr.json 2> $O/r02_bench11_lokr.err; echo "lokr rc=$?"
timeout 200 python bench.py --algo locon $B > $O/r02_bench11_locon.json 2> $O/r02_bench11_locon.err; echo "locon rc=$?"
timeout 200 python bench.py --algo locon $B --no-roofline --no-defer > $O/r02_bench11_locon_nodefer.json 2> $O/r02_bench11_locon_nodefer.err; echo "locon-nodefer rc=$?"
timeout 200 python bench.py --algo locon --model sd15 $B --no-roofline > $O/r02_bench11_sd15_locon.json 2> $O/r02_bench11_sd15_locon.err; echo "sd15 rc=$?"
for f in $O/r02_bench11_*.json; do echo "== $f"; python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'], json.dumps(d.get('roofline')))"; done
for f in $O/r02_bench11_*.err; do echo "== $f"; grep -v "^$" $f | grep -v Warning | tail -3; done
