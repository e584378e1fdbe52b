#!/usr/bin/env python3
"""Instruction mix of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only): static counts per class and
per basic block, so that the serial instruction chain of a latency-bound kernel can be read off without a GPU.

  python benchmarks/asm_mix.py /tmp/capi.s 'kron3_kernelIDF16bLi2ELb0ELi3ELb0'   [--blocks]
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"): return "mfma"
    if op.startswith("v_cvt"): return "valu_cvt"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): return "vmem_load"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"): return "vmem_store"
    if op.startswith("global_atomic") or op.startswith("buffer_atomic"): return "vmem_atomic"
    if op.startswith("scratch_"): return "scratch"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    name = lines[start].split(":")[0]
    total = collections.Counter()
    blocks = []
    cur, cur_name = collections.Counter(), "entry"
    for l in lines[start + 1:]:
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            blocks.append((cur_name, cur))
            cur, cur_name = collections.Counter(), m.group(1)
            continue
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if not m or l.startswith("\t."):
            continue
        c = classify(m.group(1))
        total[c] += 1
        cur[c] += 1
    blocks.append((cur_name, cur))
    print(name)
    print("  total", sum(total.values()), dict(sorted(total.items(), key=lambda kv: -kv[1])))
    if show_blocks:
        for n, c in blocks:
            if sum(c.values()) >= 8:
                print(f"  {n:12s} {sum(c.values()):5d}", dict(sorted(c.items(), key=lambda kv: -kv[1])))


main()
