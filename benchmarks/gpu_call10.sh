#!/bin/bash
# round-2 GPU call 10: kron3 phase-by-phase last segment: parity (linear ops) + phase stamps + launch times
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 300 python -X faulthandler -m pytest tests/test_gpu_linear_ops.py tests/test_gpu_custom_ops.py tests/test_gpu_deferred_wgrad.py -m gpu -q --timeout 60 --maxfail 10 -p no:cacheprovider -k "lokr or Lokr or kron or deferred or grouped" > $O/r02_pytest10.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest10.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest10.log | head -40
cd benchmarks
L=../$O/r02_ktrace10.log; rm -f $L
for sh in "1024 1280 1280" "4096 640 640" "1024 1280 10240" "1024 5120 1280"; do
  for m in fwd bwd; do
    echo "== $sh $m trace" >> $L; timeout 60 ./ktrace $sh $m 2>&1 | tail -2 >> $L
    echo -n "== $sh $m time: " >> $L; KT_TIME=1 timeout 60 ./ktrace $sh $m 2>&1 | tail -1 >> $L
  done
done
cat $L
