#!/bin/bash
# round 6 call 24: per-shape times of the LoKr Conv2d layers (baseline for the channel-chunked patch kernel)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
timeout 600 python benchmarks/shape_times.py --algo lokr --layers conv > $O/r06_c24_shape_times_lokr_conv.log 2>&1; echo "rc=$?"
grep -v amdgpu $O/r06_c24_shape_times_lokr_conv.log | tail -45
