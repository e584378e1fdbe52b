#!/bin/bash
# round 3, call 13: XCD-packed work order of the grouped dW2 launch: parity, A/B step time, HBM traffic of the Linear workload
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -X faulthandler -m pytest tests/test_gpu_deferred_wgrad.py tests/test_gpu_stress_guard.py tests/test_gpu_lokr_lowrank.py tests/test_gpu_grad_sync.py -q -k "not locon and not loha" --timeout 300 -p no:cacheprovider --maxfail 10 > $O/r03_c13_tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|MISMATCH|OUT-OF" $O/r03_c13_tests.log | cut -c1-300 | head
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-base --no-reference"
for v in 1 0 1 0; do echo -n "LYC_DW2G_PACKED=$v: "; LYC_DW2G_PACKED=$v timeout 200 $B 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'], d['roofline']['families_ms']['kron_dw2s_grouped'])"; done
for v in 1 0; do
  WORKLOADS="lokr/sdxl/linear" LYC_DW2G_PACKED=$v timeout 600 bash benchmarks/pmc_traffic.sh > $O/r03_c13_pmc_$v.log 2>&1
  cp $O/pmc_traffic.json $O/r03_c13_pmc_traffic_packed$v.json
  python - <<PY
import json
d=json.load(open("$O/r03_c13_pmc_traffic_packed$v.json"))["workloads"]["lokr/sdxl/linear"]["lokr_dw2s"]
print("packed=$v dw2s: read %.2f GB write %.2f GB per pass" % (d["read_bytes"]/1e9, d["write_bytes"]/1e9))
PY
done
