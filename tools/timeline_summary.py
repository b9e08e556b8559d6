#!/usr/bin/env python3
"""Kernel timeline of the last replayed `enhance` graph in a `rocprofv3 --kernel-trace` CSV (single chain): busy time, idle
time between consecutive kernels, and the idle time attributed to the kernel that FOLLOWS each gap, by kernel family.
usage: timeline_summary.py <dir with *kernel_trace.csv> [launches per call]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
path = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last call = everything after the last long idle period (> 200 us) preceding >= 300 kernels
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 200_000 and len(rows) - i >= 300:
        cut = i
call = rows[cut:]


def fam(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"^llie::", "", n)
    return re.split(r"[<(]", n)[0]


busy = sum(e - s for s, e, _ in call)
wall = call[-1][1] - call[0][0]
gap_by_next, gap_by_prev, cnt = defaultdict(int), defaultdict(int), defaultdict(int)
dur = defaultdict(int)
for i, (s, e, n) in enumerate(call):
    dur[fam(n)] += e - s
    cnt[fam(n)] += 1
    if i:
        g = max(0, s - call[i - 1][1])
        gap_by_next[fam(n)] += g
        gap_by_prev[fam(call[i - 1][2])] += g
print(f"{len(call)} kernels in the last call: wall {wall / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle between kernels {(wall - busy) / 1e3:.1f} us "
      f"({100 * (wall - busy) / wall:.1f} %), mean gap {(wall - busy) / max(1, len(call) - 1) / 1e3:.2f} us")
print(f"{'kernel family':28s} {'launches':>8s} {'busy us':>10s} {'avg us':>8s} {'gap before (us)':>16s} {'avg':>6s} {'gap after (us)':>15s} {'avg':>6s}")
for k in sorted(dur, key=lambda k: -dur[k]):
    print(f"{k:28s} {cnt[k]:8d} {dur[k] / 1e3:10.1f} {dur[k] / cnt[k] / 1e3:8.2f} {gap_by_next[k] / 1e3:16.1f} {gap_by_next[k] / cnt[k] / 1e3:6.2f} "
          f"{gap_by_prev[k] / 1e3:15.1f} {gap_by_prev[k] / cnt[k] / 1e3:6.2f}")
