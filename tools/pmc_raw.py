#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc pass directory (CSV output): usage pmc_raw.py <dir> [name filter]"""
import collections, csv, glob, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_summary import demangle

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
names = demangle(list(acc.keys()))
for k, c in acc.items():
    if flt and flt not in names[k]:
        continue
    n = len(disp[k])
    print(f"{names[k]}  x{n}")
    for cn, v in sorted(c.items()):
        print(f"    {cn:32s} {v / n:16.1f}")
