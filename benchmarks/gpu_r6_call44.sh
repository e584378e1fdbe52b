#!/bin/bash
# round 6 call 44: gemm16d with the 2-D tile -> XCD mapping behind the library: LoHa parity, the LoHa step, per-kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
G16_ONLY="128x64 D3" timeout 300 benchmarks/g16bench > $O/r06_c44_g16bench_12864.log 2>&1; echo "g16bench rc=$?"; grep -c "relerr" $O/r06_c44_g16bench_12864.log; grep "attn1280\|   g16d" $O/r06_c44_g16bench_12864.log | head -8 | cut -c1-120
timeout 1500 python -m pytest tests/test_gpu_linear_ops.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_modules_golden.py tests/test_gpu_functional_api.py tests/test_gpu_deferred_wgrad.py tests/test_gpu_golden_sweep.py -m gpu -x -q -k "loha or Loha or hada" > $O/r06_c44_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_c44_tests.log
timeout 600 python bench.py --algo loha --steps 20 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base > $O/r06_c44_bench_loha.json 2> $O/r06_c44_bench_loha.err
python3 -c "
import json;d=json.loads(open('$O/r06_c44_bench_loha.json').read().strip().splitlines()[-1]);r=d.get('roofline') or {};print('loha', d['ms_per_step'], d['value'], 'frac', r.get('frac'), r.get('achieved'), r.get('families_ms'))" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kstats --output-format csv -- python $OLDPWD/bench.py --algo loha --steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > /dev/null 2>&1
f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_c44_loha_kernel_stats.csv && head -14 "$f" | cut -c1-150
