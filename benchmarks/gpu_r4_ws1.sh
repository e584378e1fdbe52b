#!/bin/bash
# round 4: the N > 1 step on ONE GPU (world_size-1 RCCL group, every bucket really reduced) in its variants, next to the plain N = 1 step
O=gpurun_out/r04f; mkdir -p $O
Q="--steps 15 --warmup 3 --no-cpu-baseline --no-reference --no-base --no-roofline --no-per-algo"
E="MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
js() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ms_per_step", j["ms_per_step"], "|", j["config"]["graph"], "|", j["config"].get("rccl_ws1"))
except Exception as e:
    print("   no result:", e)
PY
}
timeout 300 python bench.py $Q > $O/n1.json 2> $O/n1.err; echo "plain N=1 rc=$?"; js $O/n1.json
timeout 300 python bench.py $Q --force-segments > $O/n1_forced_segments.json 2> $O/n1_forced_segments.err; echo "N=1, bucket-aligned segments, no collectives rc=$?"; js $O/n1_forced_segments.json
p=29551
for leg in "ws1_bucket_segments" "ws1_8_segments --segments 8" "ws1_reduce_scatter --collective reduce_scatter"; do
  set -- $leg; n=$1; shift; p=$((p+1))
  env $E MASTER_PORT=$p timeout 300 python bench.py $Q --rccl-ws1 "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$?"; js $O/$n.json
done
