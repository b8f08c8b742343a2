#!/bin/bash
# shader-engine counters of the two-word bucket kernel (config 5's build): rocprofv3 --pmc passes of tools/probe_c5quick.py
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_front2; rm -rf $OUT; mkdir -p $OUT
i=0
for cset in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES"; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/probe_c5quick.py > $OUT/log$i.txt 2>&1 ); tail -3 $OUT/log$i.txt
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, glob, sys, os
res = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name", "")
        if "bucket2_sort" not in kn: continue
        key = kn[kn.index("k_bucket2_sort"):kn.index("k_bucket2_sort") + 36]
        e = res.setdefault(key, {})
        e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, e in res.items():
    print(k, {a: round(b / 1e6, 2) for a, b in sorted(e.items())})
PY
