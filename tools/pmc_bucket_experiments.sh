#!/bin/bash
# Round-6 experiments on k_bucket_sort_wave's LDS traffic (bft_front.hip: FBW_PACK, FBW_DBITS, FBW_TOPBALLOT, FBW_SPLIT32): the counters of the
# bucket kernels (tools/pmc_build.py) and their durations in a plain kernel trace, for the shipped kernel and for every variant library found
# in tools/microbench/v/ (built by hand: hipcc -D... -c bft_front.hip, linked with the other objects).  usage: bash tools/pmc_bucket_experiments.sh <out dir>
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-exp}; mkdir -p "$OUT"
one() {  # name, library ("" = the shipped one)
  local name=$1 lib=$2
  [ -n "$lib" ] && export BFT_GPU_LIB=$lib || unset BFT_GPU_LIB
  python3 tools/pmc_build.py "$OUT/pmc_$name.json" 2>&1 | grep -E "^k_bucket" > "$OUT/pmc_$name.txt"
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tl_$name" -- python3 "$ROOT/tools/bench_insert.py" --reserve --sample 100000 > /dev/null 2>&1 )
  python3 tools/build_timeline.py "$OUT/tl_$name" 2>/dev/null | grep -E "k_bucket|bft_scan::k_scan<unsigned long" | head -4 > "$OUT/trace_$name.txt"
  rm -rf "$OUT/tl_$name"
  python3 tools/bench_insert.py --reserve --sample 100000 2>/dev/null | tail -n 1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); b = d.get('build_breakdown_ms', d); print({k: v for k, v in b.items() if 'redone' in k or 'sort' in k}, d.get('parity'))" > "$OUT/bench_$name.txt" 2>&1
}
one shipped ""
for lib in tools/microbench/v/lib_*.so; do n=$(basename $lib .so); one ${n#lib_} $ROOT/$lib; done
for f in "$OUT"/pmc_*.txt; do echo "== $(basename $f .txt)"; cat $f; cat "$OUT/trace_$(basename $f .txt | sed 's/^pmc_//').txt" "$OUT/bench_$(basename $f .txt | sed 's/^pmc_//').txt"; done > "$OUT/summary.txt"
