#!/usr/bin/env python3
"""Shader-engine counters of the build's kernels (what bounds a kernel: issue, waits, LDS): `rocprofv3 --pmc` passes of tools/bench_insert.py
(config 3), one counter set per pass, summed per kernel name over the process.
usage: pmc_build.py <out.json> [bench_insert options ...]      e.g.  pmc_build.py gpurun_out/pmc_build.json --opt build_groups=0"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
          ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"],
          ["SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "SQ_WAVES"]]
KEEP = ("k_bucket", "k_group", "k_cs_", "k_rs_", "k_scan", "k_assign", "k_kh_", "k_msd", "k_prefix", "k_flat")


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    for cut in ("(", "<"):
        if cut in name and not name.startswith("rocprim"):
            name = name[: name.index(cut)]
    return name[:60]


def main():
    out_path = sys.argv[1]
    extra = sys.argv[2:]
    rocprof = shutil.which("rocprofv3")
    res = {}
    tmp = tempfile.mkdtemp(prefix="bft_pmcb_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for i, cset in enumerate(PASSES):
        d = os.path.join(tmp, f"p{i}")
        cmd = [rocprof, "--kernel-trace", "--pmc", *cset, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "bench_insert.py"),
               "--reserve", "--sample", "100000", *extra]
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        if r.returncode != 0:
            res.setdefault("errors", []).append(r.stdout.decode(errors="replace")[-300:])
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                kn = short(row.get("Kernel_Name", ""))
                if not any(k in kn for k in KEEP):
                    continue
                e = res.setdefault(kn, {})
                e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        if i == 0:
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    kn = short(row.get("Kernel_Name", ""))
                    if any(k in kn for k in KEEP):
                        e = res.setdefault(kn, {})
                        e["us"] = e.get("us", 0.0) + (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                        e["launches"] = e.get("launches", 0) + 1
    shutil.rmtree(tmp, ignore_errors=True)
    for kn, e in res.items():
        if isinstance(e, dict) and e.get("SQ_WAVE_CYCLES"):
            wc = e["SQ_WAVE_CYCLES"]
            e["frac_wait_any"] = round(e.get("SQ_WAIT_ANY", 0) / wc, 3)
            e["frac_wait_inst"] = round(e.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
            e["frac_active_inst"] = round(e.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    for kn, e in sorted(res.items(), key=lambda kv: -(kv[1].get("us", 0) if isinstance(kv[1], dict) else 0)):
        if isinstance(e, dict):
            print(kn, {k_: (round(v, 1) if isinstance(v, float) else v) for k_, v in e.items()})


if __name__ == "__main__":
    main()
