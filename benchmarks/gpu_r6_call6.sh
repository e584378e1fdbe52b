#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
(for sh in ffup1280 attn1280 ffdn1280; do for v in "128x128 D3" "64x64 D3"; do echo "== $sh / $v / mode 0"; bash benchmarks/g16pmc.sh $sh "$v" 0 x; done; done) > $O/r06_c6_g16_pmc.log 2>&1
cat $O/r06_c6_g16_pmc.log
