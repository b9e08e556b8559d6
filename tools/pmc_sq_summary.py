#!/usr/bin/env python3
"""Summarise a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass (CSV output) per kernel: MFMA-pipe utilisation and where the waves' cycles go.

  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * CUs * 4 SIMDs)
                   (the gfx94x MfmaUtil formula; on MI355X the GRBM counter is accumulated over the 8 XCDs -- checked
                   against the dispatch's own duration: GRBM_GUI_ACTIVE = 8 x duration x shader clock)
  parked / issue_stall / issuing = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
  (MI355X_MICROARCH.md: the three are disjoint and sum to about SQ_WAVE_CYCLES)

usage: pmc_sq_summary.py <pass_dir> <out.json> [n_cus]
"""
import collections, csv, glob, json, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_summary import demangle

d, out = sys.argv[1], sys.argv[2]
ncu = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
names = demangle(list(acc.keys()))
res = {}
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    if gui <= 0 or wave <= 0:
        continue
    res[names[k]] = {
        "launches_sampled": len(disp[k]),
        "gpu_cycles_per_launch": round(gui / 8.0 / len(disp[k])),
        "mfma_util": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * ncu * 4), 4),
        "waves_parked": round(c.get("SQ_WAIT_ANY", 0.0) / wave, 4),
        "waves_issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4),
        "waves_issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4),
    }
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["gpu_cycles_per_launch"] * kv[1]["launches_sampled"]))
json.dump({"note": "SQ / GRBM counters of one pass; see tools/pmc_sq_summary.py for the formulas", "kernels": res}, open(out, "w"), indent=1)
for k, v in list(res.items())[:12]:
    print(f"{v['mfma_util']*100:5.1f}% mfma  parked {v['waves_parked']*100:4.1f}%  stalled {v['waves_issue_stalled']*100:4.1f}%  issuing {v['waves_issuing']*100:4.1f}%  x{v['launches_sampled']:4d}  {k}")
