#!/bin/bash
# round 6 call 52: kconv column tile that fills one round of 256 CUs (NI = 3: 4 column tiles x 64 pixel tiles = 256 workgroups on the
# 32 x 32 layers with N = 160, instead of NI = 4: 192 workgroups) -- experiment build benchmarks/kcbench_nifill (-DLYC_KCONV_NI_FILL)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
{
echo "# B C H O [bwd] : shipped plan | NI fill (us per launch, 100 back-to-back launches)"
for sh in "1 1280 32 1280" "1 1280 32 1280 bwd" "1 1920 32 1280" "1 640 32 1280" "1 640 32 1280 bwd" "1 640 64 640" "1 1280 64 1280"; do
  a=$(KT_TIME=1 timeout 60 benchmarks/kcbench $sh | grep "us per" | cut -d' ' -f1)
  b=$(KT_TIME=1 timeout 60 benchmarks/kcbench_nifill $sh | grep "us per" | cut -d' ' -f1)
  echo "$sh : $a | $b"
done
} > $O/r06_c52_kcbench_ni_fill.log 2>&1; cat $O/r06_c52_kcbench_ni_fill.log
