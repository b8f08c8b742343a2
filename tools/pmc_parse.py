#!/usr/bin/env python3
"""Parse the rocprofv3 --pmc CSVs written by tools/pmc_collect.sh into the JSON bench.py reads (profiles/pmc_traffic.json)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, wl, nq, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
KERNEL = sys.argv[5] if len(sys.argv) > 5 else "k_query"  # substring of the kernel name whose last `reps` dispatches are averaged
per_counter = defaultdict(lambda: defaultdict(float))  # counter -> dispatch -> summed value
durations = []
# bft_gpu_build launches k_query on a small batch to choose its residency: only the LAST `reps` dispatches of a pass are
# the launches of the full batch
for f in glob.glob(os.path.join(out_dir, "pass*", "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if KERNEL in r.get("Kernel_Name", "")]
    keep = set(sorted({int(r["Dispatch_Id"]) for r in rows})[-reps:])
    for row in rows:
        if int(row["Dispatch_Id"]) in keep:
            per_counter[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
for f in glob.glob(os.path.join(out_dir, "pass*", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if KERNEL in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for row in rows[-reps:]:
        durations.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
avg = {c: sum(d.values()) / len(d) for c, d in per_counter.items()}
res = {
    "_comment": f"rocprofv3 --pmc passes (one counter set per pass, --kernel-trace only; tools/pmc_collect.sh) of tools/pmc_workload.py {wl} {nq} {reps}: "
                "averages per k_query launch. FETCH_SIZE/WRITE_SIZE are reported in units of 1024 B; for this random-gather pattern "
                "FETCH_SIZE*1024 == TCC_MISS_sum*64 B (compare fetch_bytes with tcc_miss_x64), i.e. the x2 correction of MI355X_MICROARCH.md "
                "for wide coalesced streams does not apply.",
    "workload": wl, "kernel": KERNEL, "queries_per_launch": nq,
    "k_query_counters_per_launch": avg,
    "k_query_us_under_pmc": [round(x, 1) for x in durations],
}
if "FETCH_SIZE" in avg:
    res["fetch_bytes"] = avg["FETCH_SIZE"] * 1024
    res["write_bytes"] = avg.get("WRITE_SIZE", 0.0) * 1024
    res["k_query_hbm_bytes_per_launch"] = int(res["fetch_bytes"] + res["write_bytes"])
if "TCC_MISS_sum" in avg:
    res["tcc_miss_x64"] = avg["TCC_MISS_sum"] * 64
    res["l2_misses_per_query"] = avg["TCC_MISS_sum"] / nq
    res["l2_requests_per_query"] = avg.get("TCC_REQ_sum", 0.0) / nq
print(json.dumps(res, indent=1))
