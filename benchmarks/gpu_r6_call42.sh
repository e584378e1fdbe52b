#!/bin/bash
# round 6 call 42: 1x1 LoCon convs through the Linear op (4-D leaves): parity, LoCon SDXL / SD1.5 / mixed preset, LoKr headline again
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_loha_conv_ops.py tests/test_gpu_modules_golden.py tests/test_gpu_custom_ops.py tests/test_gpu_deferred_wgrad.py tests/test_gpu_golden_sweep.py tests/test_gpu_functional_api.py -m gpu -x -q > $O/r06_c42_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_c42_tests.log
for cfg in "locon:--algo locon" "locon_sd15:--algo locon --model sd15" "mixed:--algo mixed --dtype fp16" "lokr:" "lokr_sd15:--model sd15"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 600 python bench.py $flags --steps 20 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > $O/r06_c42_bench_$name.json 2> $O/r06_c42_bench_$name.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c42_bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], d['value'])" 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kstats --output-format csv -- python $OLDPWD/bench.py --algo locon --model sd15 --steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > /dev/null 2>&1
f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_c42_locon_sd15_kernel_stats.csv && head -14 "$f" | cut -c1-150
