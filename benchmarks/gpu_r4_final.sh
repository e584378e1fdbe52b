#!/bin/bash
# round 4, closing measurement on ONE box: PMC traffic on the shipped library, the driver's bench line, kernel stats, the ws1 variants
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
P=r04_final
WORKLOADS="lokr/sdxl/linear lokr/sdxl/conv" timeout 700 bash benchmarks/pmc_traffic.sh > $O/${P}_pmc.log 2>&1; echo "pmc rc=$?"; tail -2 $O/${P}_pmc.log | cut -c1-200
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py > $O/${P}_bench_default.json 2> $O/${P}_bench_default.err; echo "default rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_final_bench_default.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(j["value"], j["ms_per_step"], "| frac", r["frac"], "achieved", r["achieved"], "traffic", r.get("traffic"), "| families", r["families_ms"])
print("conv", r.get("conv", {}).get("families_ms"), r.get("conv", {}).get("frac"), r.get("conv", {}).get("traffic"))
print("ref", j.get("reference_rocm_eager")); print("base", j.get("base_plus_adapter"), j.get("value_base_plus_adapter"))
print("per_algo", {k: (v.get("ms_per_step"), v.get("steps_per_s")) for k, v in j.get("per_algo", {}).items()})
print("cpu", j.get("cpu_baseline", {}).get("value"))
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kstats --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-roofline --no-per-algo > /dev/null 2>&1
f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${P}_bench_kernel_stats.csv && head -8 "$f" | cut -c1-160
cd $R && bash benchmarks/gpu_r4_ws1.sh 2>&1 | tee $O/${P}_ws1.log
