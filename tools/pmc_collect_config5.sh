#!/bin/bash
# rocprofv3 --kernel-trace --stats summary + --pmc passes (TCC traffic) for the config-5 kernels: k_branching, k_color_rows_kh (lookup and rows
# in one launch since round 6) (tools/pmc_config5.py).  Run on the GPU box from the repo root; results under gpurun_out/pmc/config5/.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc/config5; REPS=3
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python3 "$ROOT/tools/pmc_config5.py" $REPS > "$OUT/stats.log" 2>&1
python3 "$ROOT/profiles/summarize_rocpd.py" "$(find "$OUT/stats" -name '*.db' | head -1)" > "$OUT/kernel_stats.txt"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/pmc_config5.py" $REPS > "$OUT/pass$i.log" 2>&1
done
cd "$ROOT"
python3 tools/pmc_parse.py "$OUT" config5 10000000 $REPS k_branching > "$OUT/pmc_k_branching.json"
python3 tools/pmc_parse.py "$OUT" config5 4000000 $REPS k_color_rows > "$OUT/pmc_k_color_rows.json"
find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*.db" -delete
grep -E "l2_|fetch_bytes|write_bytes" "$OUT"/pmc_k_*.json
grep -E "k_branching|k_color_rows|k_row_colorsets" "$OUT/kernel_stats.txt" | grep "grid=" | head
