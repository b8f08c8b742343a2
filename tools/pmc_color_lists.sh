#!/bin/bash
# Counters of the one-launch id-list kernel (k_colors_kh) on config 4: six rocprofv3 --pmc passes of tools/bench_color_lists.py cfg4, summed per kernel
# over the process (seven launches of the kernel per pass).  usage (GPU box, repo root): bash tools/pmc_color_lists.sh > gpurun_out/pmc_color_lists.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcc; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCC_MISS_sum TCC_REQ_sum" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/tools/bench_color_lists.py" cfg4 > "$OUT/p$i.log" 2>&1
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
res = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name", "")
        if "k_colors_kh" not in kn and "k_query_kh" not in kn: continue
        key = "k_colors_kh" if "k_colors_kh" in kn else "k_query_kh"
        e = res.setdefault(key, {})
        e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        e.setdefault("_n_" + row["Counter_Name"], 0)
        e["_n_" + row["Counter_Name"]] += 1
for k, e in res.items():
    print(k, {c: (round(v), e["_n_" + c]) for c, v in e.items() if not c.startswith("_n_")})
PY
rm -rf "$OUT"  # (the raw traces are tens of MB: only the summary above leaves the box)
