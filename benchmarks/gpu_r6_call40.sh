#!/bin/bash
# round 6 call 40: kconv PIPE with incremental tap offsets + stage 1 requested behind the prologue; 1x1 LoKr convs through the Linear op
# (4-D leaf w2): parity, per-shape timing, the LoKr conv layers and the headline step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_lokr_planes.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_modules_golden.py tests/test_gpu_custom_ops.py tests/test_gpu_deferred_wgrad.py -m gpu -x -q > $O/r06_c40_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_c40_tests.log
{
echo "# benchmarks/kcbench (ktrace_conv.cpp without -DLYC_TRACE), KT_TIME=1: 100 back-to-back launches, bf16, factor 8, 3x3 stride 1"
echo "# B C H O [bwd]   pipelined (default) | serial (KT_SERIAL=1)"
for sh in "1 1280 32 1280" "1 1280 32 1280 bwd" "1 320 128 320" "1 320 128 320 bwd" "1 640 64 640" "1 640 64 640 bwd" "1 1920 64 640" "1 1920 64 640 bwd" "1 1280 64 1280" "1 640 128 640" "1 960 128 320" "1 640 128 320 bwd"; do
  a=$(KT_TIME=1 timeout 60 benchmarks/kcbench $sh | grep "us per" | cut -d' ' -f1)
  b=$(KT_TIME=1 KT_SERIAL=1 timeout 60 benchmarks/kcbench $sh | grep "us per" | cut -d' ' -f1)
  echo "$sh : $a | $b"
done
} > $O/r06_c40_kcbench.log 2>&1; cat $O/r06_c40_kcbench.log
timeout 60 benchmarks/ktrace_conv 1 1280 32 1280 > $O/r06_c40_ktrace_1280.log 2>&1; tail -1 $O/r06_c40_ktrace_1280.log | cut -c1-600
for cfg in "conv:--layers conv" "lokr:"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 600 python bench.py $flags --steps 20 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > $O/r06_c40_bench_$name.json 2> $O/r06_c40_bench_$name.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c40_bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], d['value'])" 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kstats --output-format csv -- python $OLDPWD/bench.py --layers conv --steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > /dev/null 2>&1
f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_c40_conv_kernel_stats.csv && head -12 "$f" | cut -c1-150
