#!/usr/bin/env python3
"""Hardware counters of the query kernel of the library in this tree, collected NOW: `rocprofv3 --pmc` passes (one counter set per
pass, --kernel-trace only, separate child processes) of tools/pmc_query.py on a BASELINE workload, parsed into per-launch and
per-query figures.  bench.py calls collect() after its timed region (the counters then describe the very library that was timed);
run as a script it writes the JSON under profiles/ (the fallback bench.py reads, guarded by the source hash, when no profiler is
available).

FETCH_SIZE / WRITE_SIZE are in KiB.  Calibration on known byte counts (tools/microbench/gather.hip, profiles/r03/pmc_microbench.json):
WRITE_SIZE is exact; FETCH_SIZE tallies 64 bytes per L2 -> fabric read request whatever the request moves -- exact for a gather that
reads one whole 64-byte line (block<64>: 64.0 B tallied per gather), half of the bytes of a wide coalesced stream (k_read: 0.5 GiB
tallied per GiB read; MI355X_MICROARCH.md, HBM).  hbm_bytes below = fetch tally + write tally + the second half of the coalesced
query stream (queries x B(k) / 2)."""
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum"]]


def source_hash():
    """sha256 over the kernel sources of libbft_gpu.so: identifies the library a counter file was collected on."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "bloomfiltertrie_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")) and "hosttest" not in f and f not in ("bft_index.cpp",):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def _parse(pass_dir, kernel, reps):
    """mean counter values over the last `reps` dispatches whose kernel name contains `kernel`; their durations in us"""
    vals, durs = {}, []
    for f in glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kernel in r.get("Kernel_Name", "")]
        keep = set(sorted({int(r["Dispatch_Id"]) for r in rows})[-reps:])
        acc = {}
        for r in rows:
            if int(r["Dispatch_Id"]) in keep:
                acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for c, d in acc.items():
            vals[c] = sum(d.values()) / len(d)
    for f in glob.glob(os.path.join(pass_dir, "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted((r for r in csv.DictReader(open(f)) if kernel in r.get("Kernel_Name", "")), key=lambda r: int(r["Start_Timestamp"]))
        durs += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-reps:]]
    return vals, durs


def collect(workload="cfg4", nq=125_000_000, reps=3, kernel="k_query", opts=(), timeout=240, kmer_bytes=7, keep_dir=None):
    """Run the passes; returns a dict (possibly with "error").  Never raises: a missing profiler or a hung pass is reported."""
    rocprof = shutil.which("rocprofv3")
    if not rocprof:
        return {"error": "rocprofv3 not found"}
    out = {"workload": workload, "queries_per_launch": nq, "kernel": kernel, "launches_averaged": reps, "options": list(opts),
           "lib_source_hash": source_hash(), "counters_per_launch": {}, "kernel_us_under_pmc": []}
    tmp = keep_dir or tempfile.mkdtemp(prefix="bft_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, cset in enumerate(PASSES):
            d = os.path.join(tmp, f"pass{i}")
            cmd = [rocprof, "--kernel-trace", "--pmc", *cset, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.join(ROOT, "tools", "pmc_query.py"), workload, str(nq), str(reps), *opts]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
            except subprocess.TimeoutExpired:
                out["error"] = f"pass {cset} timed out"
                break
            if r.returncode != 0:
                out["error"] = f"pass {cset} failed rc={r.returncode}: " + r.stdout.decode(errors="replace")[-300:]
                break
            vals, durs = _parse(d, kernel, reps)
            if not vals:
                out["error"] = f"pass {cset}: no dispatch of {kernel} found"
                break
            out["counters_per_launch"].update({c: round(v, 1) for c, v in vals.items()})
            out["kernel_us_under_pmc"] += [round(x, 1) for x in durs]
    finally:
        if not keep_dir:
            shutil.rmtree(tmp, ignore_errors=True)
    c = out["counters_per_launch"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["fetch_bytes_tallied"] = c["FETCH_SIZE"] * 1024
        out["write_bytes"] = c["WRITE_SIZE"] * 1024
        out["stream_correction_bytes"] = nq * kmer_bytes / 2.0
        out["hbm_bytes_per_launch"] = out["fetch_bytes_tallied"] + out["write_bytes"] + out["stream_correction_bytes"]
        out["hbm_bytes_per_query"] = round(out["hbm_bytes_per_launch"] / nq, 3)
    if "TCC_MISS_sum" in c:
        out["l2_misses_per_query"] = round(c["TCC_MISS_sum"] / nq, 4)
        out["l2_requests_per_query"] = round(c.get("TCC_REQ_sum", 0.0) / nq, 4)
        out["ea_read_requests_per_query"] = round(c.get("TCC_EA0_RDREQ_sum", 0.0) / nq, 4)
    if out["kernel_us_under_pmc"]:
        out["kernel_us_under_pmc_mean"] = round(sum(out["kernel_us_under_pmc"]) / len(out["kernel_us_under_pmc"]), 1)
    return out


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 125_000_000
    dest = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", f"pmc_{wl}.json")
    opts = sys.argv[4:]
    kb = 8 if wl.endswith("k31") else 7
    res = collect(wl, nq, 3, "k_query", opts, kmer_bytes=kb)
    os.makedirs(os.path.dirname(dest), exist_ok=True)
    json.dump(res, open(dest, "w"), indent=1)
    print(json.dumps(res))
