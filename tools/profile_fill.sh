# per-kernel times of the k-mer hash build alone (tools/probe_fill.py under rocprofv3 --kernel-trace --stats); prints the build's kernels
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_fill; rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/t" -o t -- python3 "$ROOT/tools/probe_fill.py" ${1:-27} > "$OUT/probe.json" 2> "$OUT/err.txt" )
python3 profiles/summarize_rocpd.py "$(find "$OUT/t" -name '*.db' | head -1)" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/t"
grep -i "k_kh_\|radix\|scan\|sweep\|histogram\|fill\|Name" "$OUT/kernel_stats.txt" | head -30
cat "$OUT/probe.json"
