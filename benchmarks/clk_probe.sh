cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/clk1 /tmp/clk2
LYC_DW2F_ROWS=1024 timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES GRBM_COUNT -d /tmp/clk1 --output-format csv -- python $R/benchmarks/dw2_ab.py --only 0 --eager 3 --layers 400 > /dev/null 2>&1
LYC_DW2F_ROWS=1024 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/clk2 --output-format csv -- python $R/benchmarks/dw2_ab.py --only 0 --eager 3 --layers 400 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/clk1/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'dw2f_table_kernel' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: v for k, v in agg.items()})
f = glob.glob('/tmp/clk2/**/*kernel_trace.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'dw2f_table_kernel' in r['Kernel_Name']:
        print('duration_ns', int(r['End_Timestamp']) - int(r['Start_Timestamp']))
PY
