#!/bin/bash
# round 6, closing measurement on ONE box (the shipped library): PMC traffic per workload, the driver's bench line, kernel stats per
# algorithm, MFMA-busy counters, the N > 1 step at world_size 1, host overhead, the full GPU suite.   P=r06_final[N] selects the prefix.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
P=${P:-r06_final}
sha256sum lycoris_amd/liblycoris_amd.so | cut -c1-16 > $O/${P}_lib_sha16.txt; cat $O/${P}_lib_sha16.txt
WORKLOADS="${WORKLOADS:-lokr/sdxl/linear lokr/sdxl/conv locon/sdxl/linear loha/sdxl/linear}" timeout 1200 bash benchmarks/pmc_traffic.sh > $O/${P}_pmc.log 2>&1; echo "pmc rc=$?"; tail -2 $O/${P}_pmc.log | cut -c1-200
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/pmc_traffic.json
for f in $O/pmc_traffic_*.txt; do [ -s "$f" ] && cp "$f" $O/${P}_$(basename $f); done
# matrix-core busy share of the same launches (counters + kernel trace only): profiles/pmc_mfma.json -> roofline.mfma_busy
rm -f $O/pmc_mfma.json
for WL in lokr/sdxl/linear lokr/sdxl/conv loha/sdxl/linear locon/sdxl/linear; do
  IFS=/ read -r ALGO MODEL LAYERS <<< "$WL"; TAG=$(echo $WL | tr / _)
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_mfma_$TAG
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma_$TAG -- \
    python $R/bench.py --algo $ALGO --model $MODEL --pmc-pass 1 --layers $LAYERS > $O/${P}_pmc_mfma_$TAG.log 2>&1 || echo "mfma pass $WL failed"
  cd $R
  python benchmarks/pmc_mfma.py /tmp/pmc_mfma_$TAG --json $O/pmc_mfma.json --workload $WL > $O/${P}_pmc_mfma_$TAG.txt 2>&1; tail -4 $O/${P}_pmc_mfma_$TAG.txt | cut -c1-160
done
[ -s $O/pmc_mfma.json ] && cp $O/pmc_mfma.json profiles/pmc_mfma.json
timeout 1200 python bench.py > $O/${P}_bench_default.json 2> $O/${P}_bench_default.err; echo "default rc=$?"
python - <<PY
import json
j = json.loads(open("gpurun_out/${P}_bench_default.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(j["value"], j["ms_per_step"], "| frac", r["frac"], "achieved", r["achieved"], "traffic", r.get("traffic"), r.get("traffic_over_algorithmic"), "| families", r["families_ms"])
print("conv", r.get("conv", {}).get("families_ms"), r.get("conv", {}).get("frac"), r.get("conv", {}).get("traffic"), r.get("conv", {}).get("traffic_over_algorithmic"))
print("headline", r.get("headline")); print("mfma_busy", str(r.get("mfma_busy"))[:300])
print("scalars", {k: v for k, v in j.items() if isinstance(v, (int, float, str)) and k not in ("metric", "metric_detail", "data")})
print("ref", j.get("reference_rocm_eager")); print("base", j.get("base_plus_adapter"), j.get("value_base_plus_adapter"))
print("per_algo", {k: (v.get("ms_per_step"), v.get("steps_per_s"), (v.get("roofline") or {}).get("frac")) for k, v in j.get("per_algo", {}).items()})
print("cpu", j.get("cpu_baseline", {}).get("value"))
PY
B="--no-reference --no-base --no-per-algo --no-cpu-baseline --no-roofline"
for v in "lokr:" "locon:--algo locon" "loha:--algo loha" "ia3:--algo ia3"; do
  name=${v%%:*}; flags=${v#*:}
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kstats --output-format csv -- python $R/bench.py $flags --steps 5 --warmup 2 $B > /dev/null 2>&1
  f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${P}_bench_${name}_kernel_stats.csv && echo "$name kernel stats:" && head -5 "$f" | cut -c1-170
  cd $R
done
for v in "plain:" "nosiblings:--no-siblings" "autocast:--autocast" "ws1:--rccl-ws1" "rank16:--rank 16" "locon:--algo locon" "locon_sd15:--algo locon --model sd15" "loha:--algo loha" "ia3:--algo ia3" "mixed:--algo mixed --dtype fp16"; do
  name=${v%%:*}; flags=${v#*:}
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py $B --steps 30 --warmup 5 $flags > $O/${P}_bench_$name.json 2> $O/${P}_bench_$name.err
  echo "$name rc=$? $(tail -1 $O/${P}_bench_$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>&1 | cut -c1-200)"
done 2>&1 | tee $O/${P}_variants.log
timeout 200 python benchmarks/host_overhead.py > $O/${P}_host_overhead.log 2>&1; echo "host overhead rc=$?"; [ -s $O/host_overhead.json ] && cp $O/host_overhead.json $O/${P}_host_overhead.json; grep -v amdgpu $O/${P}_host_overhead.log | tail -3 | cut -c1-400
if [ -z "${NO_TESTS:-}" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/${P}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/${P}_pytest_gpu.log | cut -c1-300
fi
