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
import sys


def demangle(names):
    """Minimal Itanium demangler for this library's kernels: _ZN4llie<len><name>[I<args>E]Ev... with
    template arguments f / DF16_ / DF16b / Li<N>E (binutils' c++filt does not know DF16_)."""
    import re
    res = {}
    for m in names:
        mm = re.match(r"_ZN4llie(\d+)", m)
        if not mm:
            res[m] = re.sub(r"\(.*", "", m.replace("void llie::", "").replace("llie::", "")).strip()
            continue
        n = int(mm.group(1))
        name = m[mm.end():mm.end() + n]
        rest = m[mm.end() + n:]
        args = []
        if rest.startswith("I"):
            i = 1
            while i < len(rest) and rest[i] != "E":
                if rest.startswith("DF16_", i):
                    args.append("_Float16"); i += 5
                elif rest.startswith("DF16b", i):
                    args.append("__bf16"); i += 5
                elif rest[i] == "f":
                    args.append("float"); i += 1
                elif rest.startswith("Li", i):
                    j = rest.index("E", i)
                    args.append(rest[i + 2:j]); i = j + 1
                else:
                    args.append("?"); i += 1
        res[m] = name + ("<" + ", ".join(args) + ">" if args else "")
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
