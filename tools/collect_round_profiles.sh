#!/bin/bash
# Everything profiles/rNN/ holds, in one run on the GPU box (from the repo root): bench line, kernel trace of the bench command,
# PMC passes of k_query (config 2 / config 4), config 5 (trace + PMC), sequences (trace + PMC), k sweep, insertion (config 3),
# perf-probe workloads, bucket sweep.  usage: bash tools/collect_round_profiles.sh <out dir under gpurun_out/>
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-final}; mkdir -p "$OUT"
strip() { grep -v "amdgpu.ids" ; }
python3 bench.py --steps 20 --warmup 5 2> "$OUT/bench.err" | strip > "$OUT/bench.json"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/bench_trace" -o t -- python3 "$ROOT/bench.py" --steps 12 --warmup 3 --no-pcie > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" )
python3 profiles/summarize_rocpd.py "$(find "$OUT/bench_trace" -name '*.db' | head -1)" > "$OUT/bench_kernel_stats.txt" 2>&1
rm -rf "$OUT/bench_trace"
bash tools/pmc_collect.sh cfg2 100000000 3 > "$OUT/pmc_cfg2.log" 2>&1; cp gpurun_out/pmc/cfg2/pmc_traffic.json "$OUT/pmc_k_query_config2.json"
bash tools/pmc_collect.sh cfg4 125000000 3 > "$OUT/pmc_cfg4.log" 2>&1; cp gpurun_out/pmc/cfg4/pmc_traffic.json "$OUT/pmc_k_query_config4.json"
bash tools/pmc_collect_config5.sh > "$OUT/pmc_config5.log" 2>&1
cp gpurun_out/pmc/config5/pmc_k_branching.json gpurun_out/pmc/config5/pmc_k_color_rows_bm.json "$OUT/"; cp gpurun_out/pmc/config5/kernel_stats.txt "$OUT/config5_kernel_stats.txt"
python3 tools/bench_config5.py --residencies 2>&1 | strip | tail -n 1 > "$OUT/config5.json"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/seq_trace" -o t -- python3 "$ROOT/tools/bench_sequences.py" > "$OUT/sequences.log" 2>&1 )
grep "^{" "$OUT/sequences.log" > "$OUT/sequences.json"; rm -f "$OUT/sequences.log"
python3 profiles/summarize_rocpd.py "$(find "$OUT/seq_trace" -name '*.db' | head -1)" > "$OUT/sequences_kernel_stats.txt" 2>&1
rm -rf "$OUT/seq_trace"
bash tools/pmc_collect_sequences.sh > "$OUT/pmc_sequences.log" 2>&1; cp gpurun_out/pmc/sequences/pmc_k_seq_walk.json "$OUT/"
python3 tools/bench_k_sweep.py 2>&1 | strip > "$OUT/k_sweep.jsonl"
python3 tools/bench_insert.py --reserve 2>&1 | strip | tail -n 1 > "$OUT/insert_config3.json"
python3 tools/bench_insert.py --reserve --k 31 2>&1 | strip | tail -n 1 > "$OUT/insert_config3_k31.json"
python3 tools/perf_probe.py --mults 1 2>&1 | strip > "$OUT/perf_probe_workloads.jsonl"
python3 tools/bench_bucket.py --workloads cfg2,cfg4 --bits 0,8 --root-direct 3 2>&1 | strip > "$OUT/bucket_sweep.jsonl"
ls -la "$OUT"
