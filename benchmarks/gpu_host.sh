#!/bin/bash
# host-side changes only (the C-ABI library is untouched): parity of the op layer + host microseconds per layer + eager step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
timeout 300 python -X faulthandler -m pytest tests/test_gpu_custom_ops.py tests/test_gpu_deferred_wgrad.py tests/test_gpu_linear_ops.py tests/test_gpu_autocast.py tests/test_gpu_modules_golden.py -m gpu -q --timeout 60 --maxfail 10 -p no:cacheprovider > $O/r02_pytest_host.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/r02_pytest_host.log | head -20
timeout 200 python benchmarks/host_overhead.py > $O/r02_host2.log 2>&1; echo "host rc=$?"; cat $O/host_overhead.json | tr -d '\n '; echo
timeout 200 python bench.py --steps 10 --warmup 3 --eager --no-cpu-baseline > $O/r02_host_lokr_eager.json 2> $O/h.err; echo "eager rc=$? $(python -c "import json;print(json.load(open('$O/r02_host_lokr_eager.json'))['ms_per_step'])")"
