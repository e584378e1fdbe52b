#!/bin/bash
# The round's closing measurement on ONE box with the shipped library: full parity suite, PMC traffic of this build (read by
# bench.py's roofline.traffic), the default bench line, every other bench leg, host overhead, rocprofv3 kernel stats, the N = 2
# control-flow run.  Outputs under gpurun_out/; the summaries are then copied to profiles/ and committed.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R="$(pwd)"
O=$R/gpurun_out
TAG=${TAG:-r02_final}
mkdir -p $O
timeout 700 python -X faulthandler -m pytest tests -m gpu -q --timeout 90 --maxfail 25 -p no:cacheprovider > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/${TAG}_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/${TAG}_pytest.log | head -40
# PMC traffic of THIS build -> profiles/pmc_traffic.json (bench.py reads it when the library sha matches)
ALGOS="lokr locon" timeout 600 bash benchmarks/pmc_traffic.sh > $O/${TAG}_pmc.log 2>&1; echo "pmc rc=$?"
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json 2>/dev/null
cd $R
run() { n=$1; shift; timeout 400 python bench.py "$@" > $O/${TAG}_bench_$n.json 2> $O/${TAG}_bench_$n.err; echo "$n rc=$? $(python -c "import json;print(json.load(open('$O/${TAG}_bench_$n.json'))['ms_per_step'])" 2>/dev/null)"; }
run lokr --steps 20 --warmup 5
Q="--steps 10 --warmup 3 --no-cpu-baseline"
run lokr_eager $Q --eager
run lokr_nodefer $Q --no-defer --no-reference --no-base --no-roofline
run locon $Q --algo locon
run locon_nodefer $Q --algo locon --no-defer --no-reference --no-base --no-roofline
run sd15_locon $Q --algo locon --model sd15
run loha --steps 5 --warmup 2 --no-cpu-baseline --algo loha
run loha_nodefer --steps 5 --warmup 2 --no-cpu-baseline --algo loha --no-defer --no-reference --no-base --no-roofline
run ia3 $Q --algo ia3
run mixed_fp16 $Q --preset mixed --dtype fp16
timeout 200 python benchmarks/host_overhead.py > $O/${TAG}_host_overhead.log 2>&1; echo "host rc=$?"; cp $O/host_overhead.json $O/${TAG}_host_overhead.json 2>/dev/null
# per-kernel totals of the same commands (kernel trace only, no counters)
export TMPDIR=/tmp
for a in lokr locon loha; do
  (cd /tmp && rm -rf /tmp/kt_$a && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$a --output-format csv -- python $R/bench.py --algo $a --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-reference --no-base > $O/${TAG}_prof_$a.log 2>&1)
  f=$(find /tmp/kt_$a -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_bench_${a}_kernel_stats.csv
  echo "prof $a: $f"
done
# N = 2 control flow (two ranks share the one GPU, gloo): not a scaling number
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo > $O/${TAG}_n2_gloo.log 2>&1; echo "n2 rc=$?"; tail -1 $O/${TAG}_n2_gloo.log | cut -c1-200
for n in lokr locon sd15_locon loha ia3 mixed_fp16 lokr_eager; do echo "== $n"; python -c "
import json;d=json.load(open('$O/${TAG}_bench_$n.json'))
print(d['ms_per_step'], d['value']); r=d.get('roofline') or {}
print({k:r.get(k) for k in ('achieved','frac','hot_path_gbs','hot_path_frac','traffic','avg_launch_us','families_ms')})
print(d.get('reference_rocm_eager')); print(d.get('base_plus_adapter')); print(d.get('cpu_baseline'))"; done
