#!/bin/bash
# HBM bytes per launch of the step's nn.Linear launches, for bench.py's roofline.traffic (run on the GPU box):
#     bash benchmarks/pmc_traffic.sh            -> gpurun_out/pmc_traffic.json + gpurun_out/pmc_traffic_<algo>.txt
# Two SEPARATE rocprofv3 passes per algorithm (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2:
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), counters only -- no trace domains in the same run.  The eager passes of
# bench.py are profiled because counters cannot be sampled inside hipGraph replays.  Copy the outputs to profiles/.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS=""
for ALGO in ${ALGOS:-lokr locon}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_${ALGO}_$C
    rm -rf "$D"
    timeout 900 rocprofv3 --pmc $C -d "$D" --output-format csv -- \
      python "$ROOT/bench.py" --algo $ALGO --pmc-pass 1 --layers linear > "$OUT/pmc_${ALGO}_$C.log" 2>&1 || echo "pass $ALGO $C failed"
  done
  python "$ROOT/benchmarks/pmc_summary.py" /tmp/pmc_${ALGO}_FETCH_SIZE /tmp/pmc_${ALGO}_WRITE_SIZE > "$OUT/pmc_traffic_${ALGO}.txt" 2>&1
  ARGS="$ARGS $ALGO /tmp/pmc_${ALGO}_FETCH_SIZE /tmp/pmc_${ALGO}_WRITE_SIZE"
done
python "$ROOT/benchmarks/pmc_summary.py" --json "$OUT/pmc_traffic.json" $ARGS
