#!/bin/bash
# rocprofv3 --pmc passes for k_query on the config-2 workload (one counter set per pass, kernel trace only),
# then tools/pmc_parse.py -> gpurun_out/pmc/pmc_traffic.json.  Run on the GPU box from the repo root:
#   bash tools/pmc_collect.sh [workload] [queries] [reps]
set -u
WL=${1:-cfg2}; NQ=${2:-100000000}; REPS=${3:-3}; shift 3 2>/dev/null || true; OPTS="$*"   # further arguments: option=value pairs for tools/pmc_query.py
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc
OUT=$OUT/$WL$(echo "$OPTS" | tr -d ' =_')
rm -rf "$OUT"; mkdir -p "$OUT"   # the parser reads every CSV below it
cd /tmp && export TMPDIR=/tmp
i=0
# FETCH_SIZE and WRITE_SIZE do not fit one pass ("exceeds the capabilities of the hardware", and rocprofv3 then hangs):
# one derived counter per pass, every pass under its own timeout.
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum" "TCC_MISS_sum" "TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/pmc_query.py" "$WL" "$NQ" "$REPS" $OPTS > "$OUT/pass$i.log" 2>&1
done
cd "$ROOT" && python3 tools/pmc_parse.py "$OUT" "$WL" "$NQ" "$REPS" > "$OUT/pmc_traffic.json"
find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*.db" -delete
grep -E "l2_|fetch_bytes|write_bytes|k_query_us" "$OUT/pmc_traffic.json" | head -8
