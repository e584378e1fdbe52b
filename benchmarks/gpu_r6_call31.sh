#!/bin/bash
# round 6 call 31: bneck4_kernel behind the library's rank-r entry points: every GPU test that touches LoCon, then the LoCon steps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
F=$(grep -li "locon" tests/test_gpu_*.py | tr '\n' ' ')
timeout 1500 python -m pytest $F -m gpu -x -q > $O/r06_c31_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r06_c31_tests.log
for cfg in "locon:" "locon_sd15:--model sd15" "mixed:--algo mixed --dtype fp16"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  a="--algo locon"; [ "$name" = mixed ] && a=""
  timeout 600 python bench.py $a $flags --steps 20 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > $O/r06_c31_bench_$name.json 2> $O/r06_c31_bench_$name.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c31_bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], d['value'])" 2>&1 | tail -1
done
