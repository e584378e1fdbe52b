#!/bin/bash
# round-2 GPU call 5: fused LoKr backward (parity + A/B), Tucker forms
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 400 python -X faulthandler -m pytest tests/test_gpu_linear_ops.py tests/test_gpu_modules_golden.py tests/test_gpu_fullsize_properties.py tests/test_gpu_custom_ops.py "tests/test_gpu_fullsize_oracle.py::test_lokr_linear_fullsize" tests/test_gpu_functional_api.py tests/test_gpu_wspace.py -m gpu -q --timeout 90 --maxfail 25 -p no:cacheprovider > $O/r02_pytest5.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest5.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest5.log | head -60
for v in 1 0 1 0; do
LYC_FUSED_BWD=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference --no-base > $O/r02_bench5_fused$v.json 2> $O/r02_bench5_fused$v.err; echo "fused=$v rc=$?"
python -c "import json;d=json.load(open('$O/r02_bench5_fused$v.json'));print('fused=$v', d['ms_per_step'], d['roofline']['families_ms'], d['roofline']['frac'])"
done
