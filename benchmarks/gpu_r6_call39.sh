#!/bin/bash
# round 6 call 39: the software-pipelined k loop of the patch kernel (kconv PIPE): parity (pipelined / serial / 4 waves), per-shape timing
# against the serial loop, the LoKr conv layers and the headline step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_lokr_planes.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_modules_golden.py -m gpu -x -q > $O/r06_c39_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_c39_tests.log
{
echo "# benchmarks/kcbench (ktrace_conv.cpp without -DLYC_TRACE), KT_TIME=1: 100 back-to-back launches, bf16, factor 8, 3x3 stride 1"
echo "# B C H O [bwd]   pipelined (default) | serial (KT_SERIAL=1) | 4 waves (KT_W4=1)"
for sh in "1 1280 32 1280" "1 1280 32 1280 bwd" "1 320 128 320" "1 320 128 320 bwd" "1 640 64 640" "1 640 64 640 bwd" "1 1920 64 640" "1 1920 64 640 bwd" "1 1280 64 1280" "1 640 128 640" "1 960 128 320" "1 640 128 320 bwd"; do
  a=$(KT_TIME=1 timeout 60 benchmarks/kcbench $sh | grep "us per" | cut -d' ' -f1)
  b=$(KT_TIME=1 KT_SERIAL=1 timeout 60 benchmarks/kcbench $sh | grep "us per" | cut -d' ' -f1)
  c=$(KT_TIME=1 KT_W4=1 timeout 60 benchmarks/kcbench $sh | grep "us per" | cut -d' ' -f1)
  echo "$sh : $a | $b | $c"
done
echo "# row-tile pins on the pipelined kernel (KT_MI): 320@128 and 640@64"
for sh in "1 320 128 320" "1 320 128 320 bwd" "1 640 64 640" "1 640 64 640 bwd"; do
  for mi in 2 4 8; do echo "$sh MI=$mi : $(KT_TIME=1 KT_MI=$mi timeout 60 benchmarks/kcbench $sh | grep 'us per' | cut -d' ' -f1)"; done
done
} > $O/r06_c39_kcbench.log 2>&1; cat $O/r06_c39_kcbench.log
timeout 60 benchmarks/ktrace_conv 1 1280 32 1280 > $O/r06_c39_ktrace_1280.log 2>&1; tail -2 $O/r06_c39_ktrace_1280.log | cut -c1-600
for cfg in "conv:--layers conv" "lokr:"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 600 python bench.py $flags --steps 20 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > $O/r06_c39_bench_$name.json 2> $O/r06_c39_bench_$name.err
  python3 -c "
import json;d=json.loads(open('$O/r06_c39_bench_$name.json').read().strip().splitlines()[-1]);print('$name', d['ms_per_step'], d['value'])" 2>&1 | tail -1
done
