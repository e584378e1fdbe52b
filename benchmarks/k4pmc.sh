#!/bin/bash
# PMC view of one kron4 / kron4r variant on one shape (eager launches: rocprofv3 cannot sample inside graph replays)
#   benchmarks/k4pmc.sh <shape filter> <variant substring> <mode 0|1|2> <out tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_LEVEL_WAVES"; do
  out=/tmp/pmc_$4_$(echo $set | cut -c4-12)
  rm -rf $out
  K4_ONLY="$2" K4_MODE=$3 timeout 200 rocprofv3 --pmc $set -d $out --output-format csv -- $R/benchmarks/k4bench $1 --eager > /dev/null 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$set" <<'PY'
import csv, sys, collections
f, names = sys.argv[1], sys.argv[2].split()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == names[0]: cnt[k] += 1
for k, d in agg.items():
    if "kron4" not in k and "kron3" not in k: continue
    n = max(cnt[k], 1)
    print(k, "launches", n, " ".join(f"{c[3:]}={d[c]/n:.3g}" for c in names))
PY
done
