#!/bin/bash
# A/B in one box: Conv2d activations NCHW vs channels_last (conv layers only, and SD1.5 LoCon whole step)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-reference --no-base --no-roofline"
timeout 200 python bench.py $B --layers conv > $O/r02_cl_lokr_conv_nchw.json 2> $O/cl.err; echo "rc=$?"
timeout 200 python bench.py $B --layers conv --channels-last > $O/r02_cl_lokr_conv_cl.json 2> $O/cl.err; echo "rc=$?"
timeout 200 python bench.py $B --algo locon --model sd15 > $O/r02_cl_sd15_locon_nchw.json 2> $O/cl.err; echo "rc=$?"
timeout 200 python bench.py $B --algo locon --model sd15 --channels-last > $O/r02_cl_sd15_locon_cl.json 2> $O/cl.err; echo "rc=$?"
for f in $O/r02_cl_*.json; do echo "$f $(python -c "import json;print(json.load(open('$f'))['ms_per_step'])")"; done
