#!/bin/bash
# HBM bytes of the step's launches, per WORKLOAD, for bench.py's roofline.traffic (run on the GPU box):
#     bash benchmarks/pmc_traffic.sh            -> gpurun_out/pmc_traffic.json + gpurun_out/pmc_traffic_<workload>.txt
# Two SEPARATE rocprofv3 passes per workload (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2:
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), counters only -- no trace domains in the same run.  The eager passes of
# bench.py are profiled because counters cannot be sampled inside hipGraph replays.  Copy the outputs to profiles/.
# WORKLOADS: "algo/model/layers[/flag]" triples; the conv pass runs with channels_last activations (the documented default).
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS=""
for WL in ${WORKLOADS:-lokr/sdxl/linear lokr/sdxl/conv locon/sdxl/linear}; do
  IFS=/ read -r ALGO MODEL LAYERS <<< "$WL"
  TAG=$(echo $WL | tr / _)
  EXTRA=""
  [ "$LAYERS" = "conv" ] && EXTRA="--channels-last"
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_${TAG}_$C
    rm -rf "$D"
    timeout 900 rocprofv3 --pmc $C -d "$D" --output-format csv -- \
      python "$ROOT/bench.py" --algo $ALGO --model $MODEL --pmc-pass 1 --layers $LAYERS $EXTRA > "$OUT/pmc_${TAG}_$C.log" 2>&1 || echo "pass $WL $C failed"
  done
  python "$ROOT/benchmarks/pmc_summary.py" /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE > "$OUT/pmc_traffic_${TAG}.txt" 2>&1
  ARGS="$ARGS $WL /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE"
done
python "$ROOT/benchmarks/pmc_summary.py" --json "$OUT/pmc_traffic.json" $ARGS
