# per-kernel times of tools/bench_config5.py (rocprofv3 --kernel-trace --stats); prints the colour-row, branching and presence kernels
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_cfg5; rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/t" -o t -- python3 "$ROOT/tools/bench_config5.py" "$@" > "$OUT/bench.json" 2> "$OUT/err.txt" )
python3 profiles/summarize_rocpd.py "$(find "$OUT/t" -name '*.db' | head -1)" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/t"
grep "k_color_rows\|k_branching_kh\|k_query_kh\|k_row_colorsets" "$OUT/kernel_stats.txt" | grep "avg_us" | cut -c1-60,150-400
tail -1 "$OUT/bench.json" | cut -c1-300
