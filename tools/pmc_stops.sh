#!/bin/bash
# L2 requests / misses of k_query with the walk truncated after stage $1 (see tools/perf_probe.py --stops)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_stop$1; mkdir -p "$OUT"
export BFT_DEBUG_STOP=$1
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCC_MISS_sum" "TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/pmc_workload.py" cfg2 100000000 2 > "$OUT/pass$i.log" 2>&1
done
cd "$ROOT" && python3 tools/pmc_parse.py "$OUT" cfg2 100000000 2 | grep -E "per_query|TCC"
