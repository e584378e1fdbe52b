#!/bin/bash
# round 3, call 14: MFMA utilisation counters per workload (north_star: "MFMA utilisation reported against gfx950 peak"), LDS conflict
# counters of the Conv2d kernels; the mixed-preset leg after the channels_last fix
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for leg in "lokr_linear --algo lokr --layers linear" "lokr_conv --algo lokr --layers conv" "loha_linear --algo loha --layers linear" "locon_linear --algo locon --layers linear"; do
  set -- $leg; n=$1; shift
  D=/tmp/mfma_$n; rm -rf $D
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $D -- \
    python $R/bench.py "$@" --pmc-pass 1 > $O/r03_c14_mfma_$n.log 2>&1 || echo "mfma pass $n failed"
  python $R/benchmarks/pmc_mfma.py $D > $O/r03_final_pmc_mfma_$n.txt 2>&1; echo "== $n"; head -12 $O/r03_final_pmc_mfma_$n.txt | cut -c1-130
done
D=/tmp/lds_conv; rm -rf $D
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $D -- \
  python $R/bench.py --algo lokr --layers conv --pmc-pass 1 > $O/r03_c14_lds_conv.log 2>&1 || echo "lds pass failed"
python - <<'PY' > $O/r03_final_pmc_lds_lokr_conv.txt 2>&1
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/lds_conv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    if "lyc" not in k: continue
    n = max(len(v) for v in c.values())
    print(k.split("lyc")[-1][:70], n, {kk: round(sum(v) / len(v)) for kk, v in c.items()})
PY
cat $O/r03_final_pmc_lds_lokr_conv.txt | cut -c1-220 | head -14
cd $R
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --algo mixed --dtype fp16 > $O/r03_final_bench_mixed_fp16.json 2> $O/r03_final_bench_mixed_fp16.err; python -c "
import json;d=json.loads(open('$O/r03_final_bench_mixed_fp16.json').read().strip().splitlines()[-1]);print('mixed', d['ms_per_step'], d['roofline'].get('frac'))"
