#!/bin/bash
# round-2 GPU call 4: LoHa 16-bit-MFMA kernels, torch.compile, DoRA 16-bit, eager / ia3 / loha / mixed benches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 400 python -X faulthandler -m pytest tests/test_gpu_loha_conv_ops.py tests/test_gpu_custom_ops.py tests/test_gpu_modules_golden.py tests/test_gpu_wspace.py tests/test_gpu_functional_api.py tests/test_gpu_autocast.py "tests/test_gpu_fullsize_oracle.py::test_loha_linear_fullsize" "tests/test_gpu_fullsize_oracle.py::test_loha_conv2d_fullsize" -m gpu -q --timeout 90 --maxfail 25 -p no:cacheprovider > $O/r02_pytest4.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest4.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest4.log | head -60
timeout 120 python benchmarks/host_overhead.py > $O/r02_host_overhead.log 2>&1; echo "host rc=$?"; tail -3 $O/r02_host_overhead.log
timeout 200 python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench4_lokr_eager.json 2> $O/r02_bench4_lokr_eager.err; echo "eager rc=$?"
timeout 200 python bench.py --algo ia3 --steps 10 --warmup 3 > $O/r02_bench4_ia3.json 2> $O/r02_bench4_ia3.err; echo "ia3 rc=$?"
timeout 300 python bench.py --algo loha --steps 5 --warmup 2 --no-cpu-baseline --no-base > $O/r02_bench4_loha.json 2> $O/r02_bench4_loha.err; echo "loha rc=$?"
timeout 200 python bench.py --preset mixed --dtype fp16 --steps 10 --warmup 3 > $O/r02_bench4_mixed.json 2> $O/r02_bench4_mixed.err; echo "mixed rc=$?"
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_loha --output-format csv -- python $OLDPWD/bench.py --algo loha --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-reference --no-base > $OLDPWD/$O/r02_prof_loha.log 2>&1; cd $OLDPWD
cp $(find /tmp/prof_loha -name "*kernel_stats.csv" | head -1) $O/r02_v1_bench_loha_kernel_stats.csv 2>/dev/null
head -12 $O/r02_v1_bench_loha_kernel_stats.csv | cut -c1-200
for f in $O/r02_bench4_*.json; do echo "== $f"; head -c 2300 $f; echo; done
for f in $O/r02_bench4_*.err; do echo "== $f"; grep -v "^$" $f | grep -v Warning | tail -3; done
