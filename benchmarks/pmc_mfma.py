"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (development / evidence tool).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES \
        --kernel-trace --output-format csv -d <dir> -- python bench.py --pmc-pass 2 --layers linear
    python benchmarks/pmc_mfma.py <dir> [--json out.json --workload algo/model/layers]     (the json feeds bench.py: roofline.mfma_busy)

SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs of the chip, so
    MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * kernel duration * clock).
On this stack GRBM_GUI_ACTIVE comes back summed over several instances and includes the per-dispatch profiling
overhead of these 10-30 us kernels, so the gfx94x derived-counter formula (busy / (GUI_ACTIVE * CUs * 4)) is off by an
order of magnitude; the denominator used here is the dispatch duration from the same trace at a nominal 2.1 GHz (the
ratio GUI_ACTIVE / 16 / duration of the shortest kernels)."""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
CLK = 2.1  # GHz, nominal
print(f"{'kernel':64s} {'n':>6s} {'dur us':>8s} {'MFMA busy cyc':>14s} {'MfmaUtil %':>10s}")
rows = {}
for k, c in sorted(acc.items()):
    if "lyc" not in k or not dur.get(k):
        continue
    n = len(dur[k])
    d_ns = sum(dur[k]) / n
    mf = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / n
    name = k.split("lyc")[-1][:60]
    print(f"{name:64s} {n:6d} {d_ns / 1e3:8.2f} {mf:14.0f} {100.0 * mf / (1024 * d_ns * CLK):10.2f}")
    rows[k] = {"launches": n, "avg_us": round(d_ns / 1e3, 2), "mfma_busy_cycles": round(mf), "mfma_util": round(mf / (1024 * d_ns * CLK), 4),
               "total_us": round(sum(dur[k]) / 1e3, 1)}
if "--json" in sys.argv:
    import hashlib
    import json
    out = sys.argv[sys.argv.index("--json") + 1]
    wl = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "unknown"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256(open(os.path.join(root, "lycoris_amd", "liblycoris_amd.so"), "rb").read()).hexdigest()[:16]
    rec = {}
    if os.path.exists(out):
        try:
            rec = json.load(open(out))
        except ValueError:
            rec = {}
    if rec.get("lib_sha16") != h:
        rec = {"lib_sha16": h, "workloads": {}, "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace over `bench.py "
               "--pmc-pass 2` (benchmarks/pmc_mfma.py): mfma_util = busy cycles summed over the chip's 1024 SIMDs / (1024 x dispatch duration x "
               "2.1 GHz nominal)"}
    tot = sum(r["total_us"] for r in rows.values()) or 1.0
    busy = sum(r["mfma_util"] * r["total_us"] for r in rows.values()) / tot
    rec["workloads"][wl] = {"kernels": rows, "time_weighted_mfma_util": round(busy, 4)}
    json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
