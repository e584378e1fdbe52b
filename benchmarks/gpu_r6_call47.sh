#!/bin/bash
# round 6 call 47: plan constants of the implicit-Conv2d factor-gradient launch (lowrank_tn_kernel<.., GAT>): wave target / atomic budget
# variants of the library (benchmarks/experiments/lib_variants/*.so, built with -DLYC_TN_WAVE_TARGET_GAT / -DLYC_TN_ATOMIC_BUDGET_GAT)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out; mkdir -p $O
cp lycoris_amd/liblycoris_amd.so /tmp/lib_base.so
B="--steps 20 --warmup 3 --no-cpu-baseline --no-reference --no-per-algo --no-base --no-roofline"
for v in base lib_tn_w2800 lib_tn_w1400_a3x lib_tn_w6000_a3x; do
  if [ $v = base ]; then cp /tmp/lib_base.so lycoris_amd/liblycoris_amd.so; else cp benchmarks/experiments/lib_variants/$v.so lycoris_amd/liblycoris_amd.so; fi
  for cfg in "sd15_conv:--algo locon --model sd15 --layers conv" "sd15:--algo locon --model sd15" "sdxl:--algo locon"; do
    name=${cfg%%:*}; flags=${cfg#*:}
    r=$(timeout 300 python bench.py $flags $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>&1 | tail -1)
    echo "$v $name $r"
  done
done 2>&1 | tee $O/r06_c47_locon_conv_wgrad_plan.log
cp /tmp/lib_base.so lycoris_amd/liblycoris_amd.so
