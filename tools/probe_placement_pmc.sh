#!/bin/bash
# Counters of the presence kernel per dispatch, in both placement regimes of one process (tools/probe_placement_workload.py): one
# rocprofv3 pass per counter set (kernel trace + counters only), then tools/probe_placement_parse.py.  usage (GPU box, repo root):
#   bash tools/probe_placement_pmc.sh <out dir under gpurun_out/> [file with one counter set per line]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-placement_pmc}; mkdir -p "$OUT"
export BFT_GPU_POOL_MAX_MB=0
i=0
SETS=${2:-tools/probe_placement_sets.txt}
while IFS= read -r cset; do
  [ -z "$cset" ] && continue
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/probe_placement_workload.py" 6 > "$OUT/pass$i.log" 2>&1 < /dev/null )
  echo "pass $i ($cset): rc=$?"
  i=$((i+1))
done < "$SETS"
python3 tools/probe_placement_parse.py "$OUT" > "$OUT/per_dispatch.jsonl"
find "$OUT" -name '*.csv' -size +2M -delete
wc -l "$OUT/per_dispatch.jsonl"
