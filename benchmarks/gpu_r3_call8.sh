#!/bin/bash
# round 3, call 8: the low-rank LoKr path (lokr_linear_lr), the segmented-capture crash by layer family, forced segments per algo
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -X faulthandler -m pytest tests/test_gpu_lokr_lowrank.py tests/test_gpu_lokr_planes.py tests/test_gpu_deferred_wgrad.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_stress_guard.py -q -k "lowrank or lokr_lr or planes or deferred or low_rank or chain or op_ or off_fast or base_is or second_step" --timeout 300 -p no:cacheprovider --maxfail 15 > $O/r03_c8_lowrank_tests.log 2>&1; echo "lowrank tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|MISMATCH|OUT-OF" $O/r03_c8_lowrank_tests.log | cut -c1-300 | head -20
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-base --channels-last --rank 16 > $O/r03_c8_lokr_rank16.json 2> $O/r03_c8_lokr_rank16.err; echo "lokr rank16 rc=$? $(python -c "import json;d=json.load(open('$O/r03_c8_lokr_rank16.json'));print(d['ms_per_step'], d['config']['adapter_params'], d.get('reference_rocm_eager'))")"; tail -3 $O/r03_c8_lokr_rank16.err | cut -c1-300
E="MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
for v in "lokr_only --no-pg --lokr-only --skip-eager" "locon_only --no-pg --locon-only --skip-eager" "lokr_only_pg --lokr-only"; do set -- $v; n=$1; shift
  env $E MASTER_PORT=295$((RANDOM % 90 + 10)) timeout 120 python -X faulthandler benchmarks/rccl_ws1_check.py "$@" > $O/r03_c8_rccl_$n.log 2>&1; echo "rccl $n rc=$? $(grep -E 'ok|captured|eager:' $O/r03_c8_rccl_$n.log | tail -3 | tr '\n' ' ' | cut -c1-260)"; grep -E "File|Segmentation|Fatal" $O/r03_c8_rccl_$n.log | head -8 | cut -c1-200
done
for a in locon mixed; do
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-base --no-reference --no-roofline --channels-last --algo $a --force-segments --segments 6 > $O/r03_c8_${a}_segments.json 2> $O/r03_c8_${a}_segments.err; echo "$a forced segments rc=$? $(python -c "import json;d=json.load(open('$O/r03_c8_${a}_segments.json'));print(d['ms_per_step'], d['config']['graph'])")"; tail -3 $O/r03_c8_${a}_segments.err | cut -c1-200
done
