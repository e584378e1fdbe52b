#!/bin/bash
# round 5, closing measurement on ONE box (the shipped library): PMC traffic per workload, the driver's bench line, kernel stats,
# the N > 1 step at world_size 1, the full GPU suite, the five-file order that aborted in round 4 (twice, as a regression run)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
P=r05_final2
WORKLOADS="lokr/sdxl/linear lokr/sdxl/conv locon/sdxl/linear loha/sdxl/linear" timeout 900 bash benchmarks/pmc_traffic.sh > $O/${P}_pmc.log 2>&1; echo "pmc rc=$?"; tail -2 $O/${P}_pmc.log | cut -c1-200
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py > $O/${P}_bench_default.json 2> $O/${P}_bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_final2_bench_default.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(j["value"], j["ms_per_step"], "| frac", r["frac"], "achieved", r["achieved"], "traffic", r.get("traffic"), r.get("traffic_over_algorithmic"), "| families", r["families_ms"])
print("conv", r.get("conv", {}).get("families_ms"), r.get("conv", {}).get("frac"), r.get("conv", {}).get("traffic"), r.get("conv", {}).get("traffic_over_algorithmic"))
print("ref", j.get("reference_rocm_eager")); print("base", j.get("base_plus_adapter"), j.get("value_base_plus_adapter"))
print("per_algo", {k: (v.get("ms_per_step"), v.get("steps_per_s")) for k, v in j.get("per_algo", {}).items()})
print("cpu", j.get("cpu_baseline", {}).get("value"))
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kstats --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-roofline --no-per-algo > /dev/null 2>&1
f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${P}_bench_kernel_stats.csv && head -6 "$f" | cut -c1-160
cd $R
B="--no-reference --no-base --no-per-algo --no-cpu-baseline --no-roofline --steps 30 --warmup 5"
for v in "plain:" "nosiblings:--no-siblings" "ws1:--rccl-ws1" "rank16:--rank 16" "rank16_nosiblings:--rank 16 --no-siblings"; do
  name=${v%%:*}; flags=${v#*:}
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py $B $flags > $O/${P}_bench_$name.json 2> $O/${P}_bench_$name.err
  echo "$name rc=$? $(tail -1 $O/${P}_bench_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>&1 | cut -c1-200)"
done 2>&1 | tee $O/${P}_variants.log
timeout 200 python benchmarks/host_overhead.py > $O/${P}_host_overhead.log 2>&1; echo "host overhead rc=$?"; [ -s $O/host_overhead.json ] && cp $O/host_overhead.json $O/${P}_host_overhead.json; grep -v amdgpu $O/${P}_host_overhead.log | tail -3 | cut -c1-400
timeout 1200 python -m pytest tests -m gpu -x -q > $O/${P}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/${P}_pytest_gpu.log | cut -c1-300
F="tests/test_gpu_linear_ops.py tests/test_gpu_functional_api.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_loha_conv_ops.py tests/test_gpu_modules_golden.py"
for i in 1; do
  timeout 600 python -m pytest $F -m gpu -x -q -s > $O/${P}_five_file_order_run$i.log 2>&1; echo "five-file order run $i rc=$?"; tail -1 $O/${P}_five_file_order_run$i.log | cut -c1-200
done
