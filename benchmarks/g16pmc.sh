#!/bin/bash
# PMC view of gemm16 / gemm16d variants on one shape (eager launches: rocprofv3 cannot sample inside graph replays)
#   benchmarks/g16pmc.sh <shape filter> <variant substring> <mode 0|1|2> <out tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAVES" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  out=/tmp/pmc_$4_$(echo $set | cut -c1-12 | tr ' ' '_')
  rm -rf $out
  G16_EAGER=1 G16_ONLY="$2" G16_MODE=$3 timeout 200 rocprofv3 --pmc $set -d $out --output-format csv -- $R/benchmarks/g16bench $1 > /dev/null 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no counters for: $set"; continue; }
  python3 - "$f" "$set" <<'PY'
import csv, sys, collections
f, names = sys.argv[1], sys.argv[2].split()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == names[0]: cnt[k] += 1
for k, d in agg.items():
    if "gemm16" not in k: continue
    n = max(cnt[k], 1)
    print(k, "launches", n, " ".join(f"{c}={d[c]/n:.4g}" for c in names))
PY
done
