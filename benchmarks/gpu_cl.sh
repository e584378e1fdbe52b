#!/bin/bash
# A/B in one box: Conv2d activations NCHW vs channels_last (conv layers only, and SD1.5 LoCon whole step)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 200 python -X faulthandler -m pytest tests/test_gpu_custom_ops.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_modules_golden.py -m gpu -q --timeout 60 --maxfail 10 -p no:cacheprovider > $O/r02_pytest_cl.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest_cl.log | head -20
B="--steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-base --no-roofline"
timeout 200 python bench.py $B --algo locon --model sd15 > $O/r02_cl_sd15_locon_nchw.json 2> $O/cl.err; echo "rc=$?"
timeout 200 python bench.py $B --algo locon --model sd15 --channels-last > $O/r02_cl_sd15_locon_cl.json 2> $O/cl.err; echo "rc=$?"
for f in $O/r02_cl_sd15*.json; do echo "$f $(python -c "import json;print(json.load(open('$f'))['ms_per_step'])")"; done
