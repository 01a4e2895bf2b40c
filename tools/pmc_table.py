#!/usr/bin/env python3
"""Per-executed-launch means of every counter tools/pmc_probe.sh collected for the search kernel (k_knn_ck) and the fit launch.
usage: pmc_table.py <dir> <workload>"""
import collections
import csv
import glob
import json
import sys

d, w = sys.argv[1], sys.argv[2]
out = {}
for kern in ("k_knn_ck", "k_fit_reduce"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{d}/{w}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                per[r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
    res = {}
    waves = per.get("SQ_WAVES", {})
    for name, disp in per.items():
        vals = sorted(disp.values())
        if not vals:
            continue
        big = [v for v in vals if v > 0.2 * vals[-1]] if vals[-1] > 0 else vals  # executed launches
        res[name] = round(sum(big) / max(len(big), 1), 1)
    out[kern] = res
json.dump(out, open(f"{d}/{w}_pmc_table.json", "w"), indent=1)
print(w, json.dumps(out))
