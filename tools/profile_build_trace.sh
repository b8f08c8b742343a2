#!/bin/bash
# kernel trace of one config-3 build (tools/bench_insert.py) -> gpurun_out/<name>_kernel_stats.txt.  usage: bash tools/profile_build_trace.sh <name> [bench_insert args]
set -u
ROOT=$(pwd); NAME=${1:-build}; shift || true
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/${NAME}_trace" -o t -- python3 "$ROOT/tools/bench_insert.py" "$@" > "$OUT/${NAME}.json" 2> "$OUT/${NAME}.err" )
python3 profiles/summarize_rocpd.py "$(find "$OUT/${NAME}_trace" -name '*.db' | head -1)" > "$OUT/${NAME}_kernel_stats.txt" 2>&1
rm -rf "$OUT/${NAME}_trace"
head -45 "$OUT/${NAME}_kernel_stats.txt"
