#!/bin/bash
# Calibration of the gather ceiling and of the FETCH_SIZE tally for random gathers (VERDICT r2 next #1b).  Run on the GPU box from the
# repo root: bash tools/calibrate_gather.sh  -> gpurun_out/calib/{gather_timing.jsonl, stream.jsonl, pmc_*.csv summaries, calib.json}
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/calib
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -i "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_WRREQ[A-Za-z0-9_]*\|FETCH_SIZE\|WRITE_SIZE\|TCC_MISS[A-Za-z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*\|TCP_TCC_READ_REQ[A-Za-z0-9_]*" | sort -u > "$OUT/counters_available.txt"
"$ROOT/tools/microbench/gather" 2 64 1024 8192 > "$OUT/gather_timing.jsonl" 2>&1
"$ROOT/tools/microbench/stream" > "$OUT/stream.jsonl" 2>&1
i=0
for SET in "FETCH_SIZE" "TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pass$i" -- "$ROOT/tools/microbench/gather" 64 1024 8192 > "$OUT/pass$i.log" 2>&1
done
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pass$i" -- "$ROOT/tools/microbench/stream" > "$OUT/pass$i.log" 2>&1
done
cd "$ROOT" && python3 tools/calibrate_parse.py "$OUT" > "$OUT/calib.json"
find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*.db" -delete
cat "$OUT/calib.json" | head -120
