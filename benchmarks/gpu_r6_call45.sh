#!/bin/bash
# round 6 call 45: kconv with the XCD-band tile order (a halo row is fetched by one L2): parity, per-shape timing vs call 40, conv layers, headline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_lokr_planes.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_modules_golden.py tests/test_gpu_loha_conv_ops.py -m gpu -x -q > $O/r06_c45_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_c45_tests.log
{
echo "# benchmarks/kcbench, KT_TIME=1: 100 back-to-back launches, bf16, factor 8, 3x3 stride 1; XCD-band tile order (call 40: identity order)"
for sh in "1 1280 32 1280" "1 1280 32 1280 bwd" "1 320 128 320" "1 320 128 320 bwd" "1 640 64 640" "1 640 64 640 bwd" "1 1920 64 640" "1 1920 64 640 bwd" "1 1280 64 1280" "1 640 128 640" "1 960 128 320" "1 640 128 320 bwd"; do
  a=$(KT_TIME=1 timeout 60 benchmarks/kcbench $sh | grep "us per" | cut -d' ' -f1)
  echo "$sh : $a"
done
} > $O/r06_c45_kcbench.log 2>&1; cat $O/r06_c45_kcbench.log
for cfg in "conv:--layers conv" "lokr:"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 600 python bench.py $flags --steps 20 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > $O/r06_c45_bench_$name.json 2> $O/r06_c45_bench_$name.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c45_bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], d['value'])" 2>&1 | tail -1
done
