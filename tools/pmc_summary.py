#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter CSVs (several passes) -> one JSON under profiles/.
usage: python tools/pmc_summary.py out.json "command text" kernel-substring csv [csv ...]"""
import collections, csv, json, sys
out, cmd, sub = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[4:]:
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["launches"] = max(len(v) for v in cs.values())
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy_frac_of_cu_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * d["SQ_BUSY_CU_CYCLES"]), 4)
    if "SQ_WAIT_ANY" in d and d.get("SQ_WAVE_CYCLES"):
        d["wait_any_share_of_wave_cycles"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 4)
    if "SQ_WAIT_INST_ANY" in d and d.get("SQ_WAVE_CYCLES"):
        d["wait_inst_share_of_wave_cycles"] = round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 4)
    res[k] = d
json.dump({"command": cmd, "per_launch_average": res}, open(out, "w"), indent=1)
print(json.dumps({k: {c: v[c] for c in v if "frac" in c or "share" in c or c == "launches"} for k, v in res.items()}, indent=1))
