#!/usr/bin/env python3
"""One JSON line per k_query_kh dispatch of every pass of tools/probe_placement_pmc.sh: duration (us) and the pass's counters."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
for pdir in sorted(glob.glob(os.path.join(out, "pass*"))):
    if not os.path.isdir(pdir):
        continue
    dur, ctr, inst = {}, {}, {}
    for f in glob.glob(os.path.join(pdir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_query_kh" in r.get("Kernel_Name", ""):
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for f in glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_query_kh" in r.get("Kernel_Name", ""):
                d = ctr.setdefault(r["Dispatch_Id"], {})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                inst.setdefault(r["Dispatch_Id"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if r["Dispatch_Id"] not in dur and "Start_Timestamp" in r:
                    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for i, did in enumerate(sorted(ctr, key=int)):
        print(json.dumps({"pass": os.path.basename(pdir), "launch": i, "derivation": i // 3, "us": round(dur.get(did, 0.0), 1), **{c: v for c, v in ctr[did].items()},
                          **{c + "_instances": [len(v), min(v), max(v)] for c, v in inst[did].items() if len(v) > 1}}))
