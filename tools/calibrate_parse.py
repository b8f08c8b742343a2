#!/usr/bin/env python3
"""Parse tools/calibrate_gather.sh: counter values per kernel dispatch of the microbenchmarks -> bytes tallied per gather."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
timing = [json.loads(l) for l in open(os.path.join(out, "gather_timing.jsonl")) if l.startswith("{")]
stream = [json.loads(l) for l in open(os.path.join(out, "stream.jsonl")) if l.startswith("{")]
# dispatches in launch order per pass; the gather binary launches, per table size and kernel: 1 warm-up + 3 timed launches
res = {"timing": timing, "stream": stream, "pmc": {}}
for p in sorted(glob.glob(os.path.join(out, "pass*"))):
    if not os.path.isdir(p):
        continue
    per = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values in dispatch order
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        acc = defaultdict(float)
        for r in rows:
            acc[(int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, kn, cn), v in sorted(acc.items()):
            name = re.sub(r"^void ", "", kn).split("(")[0]
            per[name][cn].append(v)
    res["pmc"][os.path.basename(p)] = {k: {c: v for c, v in d.items()} for k, d in per.items()}
print(json.dumps(res, indent=1))
