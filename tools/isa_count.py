#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing, per region between labels / barriers.
usage: isa_count.py file.s mangled_kernel_name [print]"""
import collections
import sys


def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    name = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    region = collections.Counter()
    first = start
    for i in range(start, end + 1):
        l = lines[i].strip()
        if not l or l.startswith(";"):
            continue
        if l.startswith(".LBB") or l.startswith("s_barrier") or i == end:
            if sum(region.values()) > 12:
                print(f"lines {first - start:5d}-{i - start:5d}: " + " ".join(f"{k}={v}" for k, v in sorted(region.items())) + f"  [{l.split()[0]}]")
            else:
                continue
            region = collections.Counter()
            first = i
            continue
        if l.startswith("."):
            continue
        region[kind(l.split()[0])] += 1


if __name__ == "__main__":
    main()
