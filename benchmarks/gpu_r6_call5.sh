#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
(G16_ABLATE=1 G16_MODE=0 G16_ONLY="D3" timeout 600 ./benchmarks/g16bench attn1280; G16_ABLATE=1 G16_MODE=0 G16_ONLY="D3" timeout 600 ./benchmarks/g16bench ffdn1280; G16_ABLATE=1 G16_MODE=0 G16_ONLY="D3" timeout 600 ./benchmarks/g16bench ffup1280) > $O/r06_c5_g16_ablate.log 2>&1
cat $O/r06_c5_g16_ablate.log
