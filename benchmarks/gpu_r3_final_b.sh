#!/bin/bash
# round 3, final measurement session, part b: PMC traffic per workload on the shipped library, the driver-style bench line, the A/B legs, kernel stats
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
# 3. kernel stats (rocprofv3 --kernel-trace --stats) of the same commands
export TMPDIR=/tmp
for leg in "lokr" "lokr_rank16 --rank 16" "lokr_conv --layers conv" "locon --algo locon" "loha --algo loha"; do
  set -- $leg; n=$1; shift
  (cd /tmp && rm -rf /tmp/kt_$n && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$n --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference --no-base --no-roofline "$@" > $O/${P}_prof_$n.log 2>&1)
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${P}_bench_${n}_kernel_stats.csv && echo "stats $n: $(head -4 $O/${P}_bench_${n}_kernel_stats.csv | tail -3 | cut -c1-120 | tr '\n' '|')"
done
