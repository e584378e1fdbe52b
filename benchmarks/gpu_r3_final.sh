#!/bin/bash
# round 3, final measurement session: PMC traffic per workload on the shipped library, the driver-style bench line, the A/B legs, kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
P=r03_final
js() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no json:", e); sys.exit()
r = d.get("roofline") or {}
ref = d.get("reference_rocm_eager") or {}
print(d["ms_per_step"], "ms; roofline", r.get("achieved"), r.get("unit"), "frac", r.get("frac"), "traffic", r.get("traffic"), "| conv", (r.get("conv") or {}).get("families_ms"), (r.get("conv") or {}).get("frac"),
      "| ref", ref.get("reference_eager_ms"), ref.get("native_eager_ms"), ref.get("speedup_eager_vs_eager"), ref.get("speedup_graph_vs_graph"), ref.get("speedup_native_graph_vs_reference_eager"),
      "| base+adapter", (d.get("base_plus_adapter") or {}).get("base_plus_adapter_ms"), d.get("value_base_plus_adapter"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
}
# 1. HBM traffic counters, per workload, on this build
WORKLOADS="lokr/sdxl/linear lokr/sdxl/conv locon/sdxl/linear" timeout 900 bash benchmarks/pmc_traffic.sh > $O/${P}_pmc.log 2>&1; echo "pmc rc=$?"; tail -3 $O/${P}_pmc.log | cut -c1-200
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/pmc_traffic.json
# 2. the driver's command (defaults)
timeout 600 python bench.py > $O/${P}_bench_default.json 2> $O/${P}_bench_default.err; echo "default rc=$?"; js $O/${P}_bench_default.json
Q="--steps 10 --warmup 3 --no-cpu-baseline"
for leg in "lokr_nchw --nchw --no-base" "lokr_noplanes --no-planes --no-reference --no-base" "lokr_nodefer --no-defer --no-reference --no-base --no-roofline" \
           "lokr_eager --eager --no-reference --no-base --no-roofline" "lokr_rank16 --rank 16 --no-base" "locon --algo locon" "locon_sd15 --algo locon --model sd15" \
           "loha --algo loha --steps 5 --warmup 2" "ia3 --algo ia3" "mixed_fp16 --algo mixed --dtype fp16" "lokr_fp16 --dtype fp16 --no-reference --no-base"; do
  set -- $leg; n=$1; shift
  timeout 400 python bench.py $Q "$@" > $O/${P}_bench_$n.json 2> $O/${P}_bench_$n.err; echo "$n rc=$?"; js $O/${P}_bench_$n.json
done
E="MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
env $E MASTER_PORT=29541 timeout 300 python bench.py $Q --no-reference --no-base --no-roofline --rccl-ws1 > $O/${P}_bench_rccl_ws1.json 2> $O/${P}_bench_rccl_ws1.err; echo "rccl-ws1 rc=$?"; js $O/${P}_bench_rccl_ws1.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_final_bench_rccl_ws1.json").read().strip().splitlines()[-1]); print("rccl_ws1:", d["config"].get("rccl_ws1"), d["config"]["graph"])
PY
timeout 200 python benchmarks/host_overhead.py > $O/${P}_host_overhead.log 2>&1; cp $O/host_overhead.json $O/${P}_host_overhead.json 2>/dev/null; tail -2 $O/${P}_host_overhead.log | cut -c1-300
timeout 100 python benchmarks/chain_bench.py > $O/${P}_chain_bench.json 2>/dev/null; cat $O/${P}_chain_bench.json
timeout 200 python benchmarks/eager_profile.py 2>&1 | grep -E "^layers" > $O/${P}_eager_host.txt; cat $O/${P}_eager_host.txt
