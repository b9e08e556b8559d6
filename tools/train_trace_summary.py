#!/usr/bin/env python3
"""One training step in a `rocprofv3 --kernel-trace` CSV of `bench.py --train`: steps are delimited by the optimiser's last
multi_tensor_apply launch; for a step in the middle of the run prints wall time, the time with at least one kernel in flight,
the summed kernel time per stream, the idle gaps (count, sum, largest) and the kernels around the largest gaps.
usage: train_trace_summary.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?")))
    rows.sort()
    opt = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[2]]
    ends = [opt[i] for i in range(len(opt)) if i + 1 == len(opt) or rows[opt[i + 1]][0] - rows[opt[i]][1] > 2_000_000]
    print(f"{len(rows)} kernels, {len(ends)} optimiser phases")
    k = len(ends) // 2
    step = rows[ends[k - 1] + 1:ends[k] + 1]
    t0, t1 = step[0][0], max(r[1] for r in step)
    print(f"step {k}: {len(step)} kernels, wall {(t1 - t0) / 1e6:.2f} ms (previous step's end to this one's: {(rows[ends[k]][1] - rows[ends[k - 1]][1]) / 1e6:.2f} ms)")
    per = defaultdict(lambda: [0, 0])
    for s, e, n, st in step:
        per[st][0] += e - s
        per[st][1] += 1
    for st, (d, n) in per.items():
        print(f"  stream {st}: {n} kernels, {d / 1e6:.2f} ms")
    busy, gaps = 0, []
    cs, ce = step[0][0], step[0][1]
    lastname = step[0][2]
    for s, e, n, st in step[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append((s - ce, lastname, n))
            cs, ce = s, e
        else:
            ce = max(ce, e)
        lastname = n
    busy += ce - cs
    print(f"  at least one kernel in flight {busy / 1e6:.2f} ms; {len(gaps)} idle gaps, {sum(g[0] for g in gaps) / 1e6:.2f} ms")
    for g, a, b in sorted(gaps, reverse=True)[:12]:
        print(f"    {g / 1e3:8.1f} us between {a[:50]} -> {b[:50]}")
    fam = defaultdict(lambda: [0, 0])
    for s, e, n, st in step:
        key = n.split("(")[0][:60]
        fam[key][0] += e - s
        fam[key][1] += 1
    for key, (d, n) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"  {key:62s} x{n:4d} {d / 1e3:9.1f} us")


if __name__ == "__main__":
    main()
