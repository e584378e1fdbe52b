#!/bin/bash
# PMC view of the grouped weight-gradient kernel on one SDXL shape (eager launches: rocprofv3 cannot sample inside graph replays)
#   benchmarks/dw2pmc.sh <shape index of dw2_ab.py> <out tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_LEVEL_WAVES"; do
  out=/tmp/pmc_$2_$(echo $set | cut -c4-12)
  rm -rf $out
  timeout 200 rocprofv3 --pmc $set -d $out --output-format csv -- python $R/benchmarks/dw2_ab.py --only $1 --eager 3 --layers ${DW2_LAYERS:-24} > /dev/null 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$set" <<'PY'
import csv, sys, collections
f, names = sys.argv[1], sys.argv[2].split()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == names[0]: cnt[k] += 1
for k, d in agg.items():
    if "dw2" not in k: continue
    n = max(cnt[k], 1)
    print(k, "launches", n, " ".join(f"{c[3:]}={d[c]/n:.4g}" for c in names))
PY
done
