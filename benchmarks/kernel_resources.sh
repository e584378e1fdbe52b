#!/bin/bash
# Register / scratch / occupancy table of every kernel in the library (compile-only; no GPU needed).
#   benchmarks/kernel_resources.sh [name filter]
cd "$(dirname "$0")/../lycoris_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c capi.hip -o /tmp/lyc_capi_res.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, subprocess, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur: rows[cur][m.group(1).split()[0]] = int(m.group(2))
names = list(rows)
# (binutils c++filt does not know DF16b / DF16_: hand it the vendor / half spellings)
dem = subprocess.run(["c++filt"], input="\n".join(n.replace("DF16b", "u6__bf16").replace("DF16_", "Dh") for n in names), capture_output=True, text=True).stdout.split("\n")
print("%5s %5s %7s %4s %7s  kernel" % ("VGPR", "AGPR", "scratch", "occ", "LDS"))
for n, d in zip(names, dem):
    d = d.replace("lyc::", "").replace("void ", "")
    d = re.sub(r"\(.*", "", d)
    if flt in d:
        r = rows[n]; print("%5d %5d %7d %4d %7d  %s" % (r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS", -1), d))
' "$1"
