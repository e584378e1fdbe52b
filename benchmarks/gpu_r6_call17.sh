#!/bin/bash
# round 6 call 17: side-stream flushes of the grouped weight gradients: tests, then A/B on the LoKr / LoHa / LoCon steps; (IA)^3 forward on the slab layout
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_deferred_wgrad.py tests/test_gpu_linear_ops.py tests/test_gpu_custom_ops.py -m gpu -x -q > $O/r06_c17_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r06_c17_tests.log
for algo in lokr loha locon; do
for flag in "" "--wgrad-side-stream"; do
  timeout 600 python bench.py --algo $algo --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline $flag > $O/r06_c17_bench_$algo$flag.json 2> $O/r06_c17_bench_$algo$flag.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c17_bench_$algo$flag.json').read().strip().splitlines()[-1]);print('$algo $flag', d['ms_per_step'])"; tail -1 $O/r06_c17_bench_$algo$flag.err | cut -c1-200
done; done
timeout 600 python bench.py --algo ia3 --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base > $O/r06_c17_bench_ia3.json 2> $O/r06_c17_bench_ia3.err
python3 -c "
import json;d=json.loads(open('$O/r06_c17_bench_ia3.json').read().strip().splitlines()[-1]);r=d.get('roofline') or {};print('ia3', d['ms_per_step'], r.get('frac'), r.get('families_ms'))"
