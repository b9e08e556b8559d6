#!/usr/bin/env python3
"""Concurrency picture of the last replayed `enhance` graph in a `rocprofv3 --kernel-trace` CSV (two half-batch branches):
how much of the wall time has 0 / 1 / 2+ kernels in flight, how much of it has ONLY latency-bound small launches in flight
(GroupNorm / Gram finalize, SE, time / FiLM, zero fill: the chip is idle for practical purposes), and each kernel family's
summed duration inside the graph.  usage: overlap_summary.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

SMALL = ("gn_finalize", "gram_finalize", "se_", "time_embed", "film", "zero_fill")


def fam(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"^llie::", "", n)
    m = re.match(r"_ZN4llie\d+([a-z0-9_]+?)I", n)
    if m:
        return m.group(1)
    return re.split(r"[<(]", n)[0]


def main():
    d = sys.argv[1]
    path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam(r["Kernel_Name"])))
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - max(e for _, e, _ in rows[max(0, i - 8):i]) > 200_000 and len(rows) - i >= 300:
            cut = i
    call = rows[cut:]
    t0, t1 = call[0][0], max(e for _, e, _ in call)
    ev = []
    for s, e, k in call:
        small = k.startswith(SMALL)
        ev.append((s, 1, small))
        ev.append((e, -1, small))
    ev.sort()
    active = big = 0
    hist = defaultdict(int)
    only_small = 0
    last = t0
    for t, dlt, small in ev:
        hist[min(active, 3)] += t - last
        if active > 0 and big == 0:
            only_small += t - last
        last = t
        active += dlt
        if not small:
            big += dlt
    wall = t1 - t0
    print(f"{len(call)} kernels in the last call: wall {wall / 1e3:.1f} us")
    for n in sorted(hist):
        print(f"  {n}{'+' if n == 3 else ' '} kernels in flight: {hist[n] / 1e3:9.1f} us ({100 * hist[n] / wall:4.1f} %)")
    print(f"  only small launches in flight: {only_small / 1e3:9.1f} us ({100 * only_small / wall:4.1f} %)")
    dur, cnt = defaultdict(int), defaultdict(int)
    for s, e, k in call:
        dur[k] += e - s
        cnt[k] += 1
    tot = sum(dur.values())
    print(f"  sum of kernel durations {tot / 1e3:.1f} us = {tot / wall:.2f} x wall")
    for k in sorted(dur, key=lambda k: -dur[k]):
        print(f"  {k:28s} x{cnt[k]:4d} {dur[k] / 1e3:10.1f} us  avg {dur[k] / cnt[k] / 1e3:8.2f}")


if __name__ == "__main__":
    main()
