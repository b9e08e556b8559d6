#!/usr/bin/env python3
"""Summarise `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (CSV output) into per-kernel HBM
traffic per launch, with the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts a
wide coalesced read stream at half its bytes, so reads = 2 x FETCH_SIZE; WRITE_SIZE is exact for 16-byte
streaming stores.  Both counters are in KiB per dispatch.

usage: pmc_summary.py <fetch_pass_dir> <write_pass_dir> <out.json> [note]
"""
import collections
import csv
import glob
import json
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    res = {}
    for m, d in zip(names, out):
        d = d.replace("void llie::", "").replace("llie::", "")
        depth, cut = 0, len(d)
        for i, ch in enumerate(d):           # drop the argument list, keep template arguments
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        res[m] = d[:cut].strip()
    return res


def collect(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    names = demangle(sorted(set(fetch) | set(write)))
    out = {}
    for m, n in names.items():
        f, w = fetch.get(m, []), write.get(m, [])
        if not f or not w or "kernel" not in n:
            continue
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        out[n] = {"launches_sampled": len(f), "fetch_size_kib_per_launch": round(fk, 1), "write_size_kib_per_launch": round(wk, 1),
                  "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    json.dump({"note": sys.argv[4] if len(sys.argv) > 4 else "", "correction": "reads = 2 x FETCH_SIZE (gfx950), writes = WRITE_SIZE; KiB",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for n, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
        print(f"{v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch  {n}")


if __name__ == "__main__":
    main()
