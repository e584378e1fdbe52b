#!/bin/bash
# round 6 call 18: one-launch plane refresh through a device table: tests, LoKr step, pack kernels in the kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lokr_planes.py tests/test_gpu_lokr_lowrank.py tests/test_gpu_siblings.py tests/test_gpu_adapted_linear.py -m gpu -x -q > $O/r06_c18_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r06_c18_tests.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base > $O/r06_c18_bench_lokr.json 2> $O/r06_c18_bench_lokr.err
python3 -c "
import json;d=json.loads(open('$O/r06_c18_bench_lokr.json').read().strip().splitlines()[-1]);r=d.get('roofline') or {};print('lokr', d['ms_per_step'], r.get('frac'), r.get('families_ms'))"
timeout 600 python bench.py --rank 16 --steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline > $O/r06_c18_bench_lokr_r16.json 2> $O/r06_c18_bench_lokr_r16.err
python3 -c "
import json;d=json.loads(open('$O/r06_c18_bench_lokr_r16.json').read().strip().splitlines()[-1]);print('lokr rank 16', d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_lokr
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_lokr --output-format csv -- python $R/bench.py --algo lokr --steps 10 --warmup 2 --no-cpu-baseline --no-reference --no-base --no-per-algo --no-roofline > $O/r06_c18_prof_lokr.json 2> $O/r06_c18_prof_lokr.err
f=$(find /tmp/prof_lokr -name "*kernel_stats.csv" | head -1); cp "$f" $O/r06_c18_lokr_kernel_stats.csv
grep -a -E "pack" $O/r06_c18_lokr_kernel_stats.csv | cut -c1-200
