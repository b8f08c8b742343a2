#!/usr/bin/env python3
"""Timeline of ONE bft_gpu_build: the kernels between the last insertion of tools/bench_insert.py and the first torch kernel after
the build, with start offsets, durations and the idle gaps between them (host overhead, allocations, synchronisations).
usage (on the GPU box):  python3 tools/build_timeline.py <dir with rocprofv3 --kernel-trace --output-format csv output>"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name[:86]


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last_pack = max(i for i, r in enumerate(rows) if "k_pack_to_tform" in r["Kernel_Name"])
    seg = []
    for r in rows[last_pack + 1:]:
        if r["Kernel_Name"].startswith("at::") or "at::native" in r["Kernel_Name"][:40]:
            break
        seg.append(r)
    t0 = int(rows[last_pack]["End_Timestamp"])
    prev_end = t0
    busy = 0
    print(f"{'start_ms':>9} {'dur_us':>9} {'gap_us':>8}  kernel")
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e3:9.1f} {(s - prev_end) / 1e3:8.1f}  {short(r['Kernel_Name'])}")
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"# kernels {len(seg)}  span {(prev_end - t0) / 1e6:.3f} ms  busy (sum of durations) {busy / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
