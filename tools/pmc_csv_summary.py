#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter_collection.csv per kernel name and counter:
   python tools/pmc_csv_summary.py <counter_collection.csv> [...] > summary.csv"""
import csv, sys, collections
agg = collections.OrderedDict()
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0]
            if "rtx::" not in k: continue
            k = k.replace("void ", "")
            d = agg.setdefault(k, {"dispatches": set(), "vgpr": r["VGPR_Count"], "lds": r["LDS_Block_Size"]})
            d["dispatches"].add((path, r["Dispatch_Id"]))
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({c for d in agg.values() for c in d if c not in ("dispatches", "vgpr", "lds")})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "vgpr", "lds_bytes", "dispatches_per_pass"] + names)
for k, d in agg.items():
    npass = len({p for p, _ in d["dispatches"]})
    w.writerow([k, d["vgpr"], d["lds"], len(d["dispatches"]) // max(npass, 1)] + ["%.6g" % d.get(c, 0.0) for c in names])
