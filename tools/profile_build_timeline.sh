#!/bin/bash
# timeline of ONE config-3 build (tools/bench_insert.py under rocprofv3 --kernel-trace, csv) -> gpurun_out/<name>_timeline.txt (tools/build_timeline.py)
set -u
ROOT=$(pwd); NAME=${1:-build}; shift || true
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/${NAME}_csv" -o t -- python3 "$ROOT/tools/bench_insert.py" "$@" > "$OUT/${NAME}.json" 2> "$OUT/${NAME}.err" )
python3 tools/build_timeline.py "$OUT/${NAME}_csv" > "$OUT/${NAME}_timeline.txt" 2>&1
rm -rf "$OUT/${NAME}_csv"
cat "$OUT/${NAME}_timeline.txt"
