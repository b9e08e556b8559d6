#!/usr/bin/env python3
"""Summarise a `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE` pass per kernel: where the waves' cycles go (parked on
s_waitcnt / barrier, issue-stalled and how much of that on the LDS queue), VALU / LDS instruction activity, and the share
of LDS-array cycles that are bank-conflict cycles.  usage: pmc_lds_summary.py <pass_dir> [top_n]"""
import collections
import csv
import glob
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_summary import demangle

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in rows:
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    n[r["Kernel_Name"]].add(r["Dispatch_Id"])
names = demangle(list(acc.keys()))
out = sorted(((c["SQ_WAVE_CYCLES"], names[k], len(n[k]), c) for k, c in acc.items() if c["SQ_WAVE_CYCLES"] > 0), key=lambda t: -t[0])
print("# share of wave cycles: parked (s_waitcnt / barrier) | issue-stalled (of which on the LDS queue) | VALU / LDS instruction active | LDS bank-conflict cycles per LDS-array cycle")
for w, k, m, c in out[:top]:
    print(f"parked {c['SQ_WAIT_ANY'] / w * 100:5.1f}%  stalled {c['SQ_WAIT_INST_ANY'] / w * 100:5.1f}% (LDS {c['SQ_WAIT_INST_LDS'] / w * 100:5.1f}%)  "
          f"valu {c['SQ_ACTIVE_INST_VALU'] / w * 100:5.1f}%  lds {c['SQ_ACTIVE_INST_LDS'] / w * 100:5.1f}%  "
          f"conflict {c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1) * 100:5.1f}%  x{m:4d}  {k[:100]}")
