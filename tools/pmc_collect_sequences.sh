#!/bin/bash
# rocprofv3 --pmc passes (TCC traffic, SQ counters) for the sequence-query kernel k_seq_walk (tools/bench_sequences.py: the last five
# dispatches are the device-resident canonical calls).  Run on the GPU box from the repo root; results under gpurun_out/pmc/sequences/.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc/sequences; REPS=5
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum" "TCC_MISS_sum" "TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_ATOMIC_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/tools/bench_sequences.py" --check 0 > "$OUT/pass$i.log" 2>&1
done
cd "$ROOT"
python3 tools/pmc_parse.py "$OUT" sequences 124000000 $REPS k_seq_walk > "$OUT/pmc_k_seq_walk.json"
find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*.db" -delete
grep -v "^  [0-9]" "$OUT/pmc_k_seq_walk.json" | cut -c1-200
