#!/bin/bash
# development helper: rebuild the library + tools, and only if that worked run a command on the MI355X box
#   benchmarks/gb.sh [gpurun timeout] 'command'
set -e
cd "$(dirname "$0")/.."
make -C lycoris_amd/csrc > /tmp/gb_build.log 2>&1 || { grep -E "error" -A5 /tmp/gb_build.log | head -40; echo "BUILD FAILED"; exit 1; }
echo "warnings: $(grep -c warning /tmp/gb_build.log)"
make -C lycoris_amd/csrc kbench ktrace > /tmp/gb_tools.log 2>&1 || { grep -E "error" -A5 /tmp/gb_tools.log | head -40; echo "TOOL BUILD FAILED"; exit 1; }
test lycoris_amd/liblycoris_amd.so -nt lycoris_amd/csrc/capi.hip || { echo "library older than sources"; exit 1; }
T=${2:+$1}; CMD=${2:-$1}
/usr/local/graft/bin/gpurun --timeout ${T:-300} -- "$CMD" 2>&1 | grep -v "^\[gpurun\]"
