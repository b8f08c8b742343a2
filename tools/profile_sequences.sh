# per-kernel times of tools/bench_sequences.py (rocprofv3 --kernel-trace --stats); prints the k_seq_* kernels
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_seq; rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/t" -o t -- python3 "$ROOT/tools/bench_sequences.py" > "$OUT/bench.json" 2> "$OUT/err.txt" )
python3 profiles/summarize_rocpd.py "$(find "$OUT/t" -name '*.db' | head -1)" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/t"
grep "k_seq" "$OUT/kernel_stats.txt" | grep "avg_us" | cut -c1-50,130-330
