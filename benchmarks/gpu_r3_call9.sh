#!/bin/bash
# round 3, call 9: bisect the capture_end crash of benchmarks/rccl_ws1_check.py; the N > 1 step with a world_size-1 RCCL group in bench.py
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
E="MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
env $E MASTER_PORT=29533 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-base --no-reference --no-roofline --channels-last --rccl-ws1 > $O/r03_c9_bench_rccl_ws1.json 2> $O/r03_c9_bench_rccl_ws1.err; echo "bench rccl-ws1 rc=$? $(python -c "import json;d=json.load(open('$O/r03_c9_bench_rccl_ws1.json'));print(d['ms_per_step'], d['config']['graph'], d['config'].get('rccl_ws1'))")"; tail -3 $O/r03_c9_bench_rccl_ws1.err | cut -c1-200
for v in "side --no-pg --skip-eager --side-stream" "notruth --no-pg --skip-eager --no-truth" "bigb --no-pg --skip-eager --big-buckets" "side_notruth_bigb --no-pg --skip-eager --side-stream --no-truth --big-buckets" "full_side --side-stream"; do set -- $v; n=$1; shift
  env $E MASTER_PORT=295$((RANDOM % 90 + 10)) timeout 120 python -X faulthandler benchmarks/rccl_ws1_check.py "$@" > $O/r03_c9_rccl_$n.log 2>&1; echo "rccl $n rc=$? $(grep -E 'ok|captured|eager:' $O/r03_c9_rccl_$n.log | tail -3 | tr '\n' ' ' | cut -c1-260)"; grep -E "File|Segmentation|Fatal|Error" $O/r03_c9_rccl_$n.log | head -6 | cut -c1-200
done
