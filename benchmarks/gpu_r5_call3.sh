#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== MIOpen tiny fp32 conv, no lycoris_amd in the process =="
timeout 400 python benchmarks/miopen_tiny_conv_repro.py 6000 > gpurun_out/r05_miopen_repro.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r05_miopen_repro.log | cut -c1-300
echo "== RCCL 1-rank collective cost =="
timeout 200 python benchmarks/rccl_ws1_probe.py > gpurun_out/r05_rccl_probe.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r05_rccl_probe.log | tail -12
echo "== rccl ws1 check =="
timeout 300 python benchmarks/rccl_ws1_check.py --comm rccl > gpurun_out/r05_rccl_ws1_check.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r05_rccl_ws1_check.log
