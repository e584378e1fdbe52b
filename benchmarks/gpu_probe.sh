#!/bin/bash
# short GPU probe: LoHa factor-gradient kernels at two waves per SIMD (lean instantiations): parity + group sweep + step A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 300 python -X faulthandler -m pytest tests/test_gpu_deferred_wgrad.py tests/test_gpu_linear_ops.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_wspace.py -m gpu -q --timeout 60 --maxfail 10 -p no:cacheprovider -k "loha or Loha or hada or small_batches" > $O/r02_pytest_probe.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest_probe.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest_probe.log | head -40
rm -f $O/r02_wgbench5.log
for v in h32 h128 h512; do
  echo "=== $v" >> $O/r02_wgbench5.log
  timeout 200 ./benchmarks/wgbench_$v 2>&1 | grep -i "loha\|rc" >> $O/r02_wgbench5.log; echo "rc=$?" >> $O/r02_wgbench5.log
done
cat $O/r02_wgbench5.log
B="--steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-roofline"
timeout 300 python bench.py --algo loha $B > $O/r02_probe_loha.json 2> $O/r02_probe_loha.err; echo "loha rc=$?"
timeout 300 python bench.py --algo loha $B --no-defer > $O/r02_probe_loha_nodefer.json 2> $O/r02_probe_loha_nodefer.err; echo "loha-nodefer rc=$?"
for f in $O/r02_probe_*.json; do echo "== $f"; python -c "import json;d=json.load(open('$f'));print(d['ms_per_step'])"; done
