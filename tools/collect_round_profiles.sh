#!/bin/bash
# Everything profiles/rNN/ holds, in one run on the GPU box (from the repo root): the bench line, the kernel trace of the bench command,
# the counter passes of the query kernel (the fallback file bench.py reads when no profiler is available), config 5, k sweep, insertion
# (with its kernel trace), sequence queries.  usage: bash tools/collect_round_profiles.sh <out dir under gpurun_out/>
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-final}; mkdir -p "$OUT"
strip() { grep -v "amdgpu.ids" ; }
python3 bench.py --steps 20 --warmup 5 2> "$OUT/bench.err" | strip > "$OUT/bench.json"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/bench_trace" -o t -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-pmc > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" )
python3 profiles/summarize_rocpd.py "$(find "$OUT/bench_trace" -name '*.db' | head -1)" > "$OUT/bench_kernel_stats.txt" 2>&1
rm -rf "$OUT/bench_trace"
python3 - "$OUT" <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
from tools import pmc_live
out = {}
for wl, nq, kb in (("cfg4", 125_000_000, 7), ("cfg4k31", 125_000_000, 8), ("cfg2", 100_000_000, 7)):
    out[wl] = pmc_live.collect(wl, nq, 3, "k_query", kmer_bytes=kb)
out["cfg4_walk_hash"] = pmc_live.collect("cfg4", 125_000_000, 3, "k_query", opts=("walk_hash=1",), kmer_bytes=7)
out["cfg4_walk"] = pmc_live.collect("cfg4", 125_000_000, 3, "k_query", opts=("kmer_hash=0",), kmer_bytes=7)
out["cfg4_walk_pure"] = pmc_live.collect("cfg4", 125_000_000, 3, "k_query", opts=("kmer_hash=0", "root_direct=0", "node_hash=0"), kmer_bytes=7)
json.dump(out, open(os.path.join(sys.argv[1], "pmc_query.json"), "w"), indent=1)
PY
python3 tools/bench_config5.py 2>&1 | strip | tail -n 1 > "$OUT/config5.json"
python3 tools/bench_k_sweep.py 2>&1 | strip > "$OUT/k_sweep.jsonl"
python3 tools/bench_insert.py --reserve --add-genome --stages --warm-pool --cpu-baseline 8 2>&1 | strip | tail -n 1 > "$OUT/insert_config3.json"
python3 tools/bench_insert.py --reserve --add-genome --stages --warm-pool --k 31 2>&1 | strip | tail -n 1 > "$OUT/insert_config3_k31.json"
python3 tools/bench_config5_build.py 2>&1 | strip | tail -n 1 > "$OUT/config5_build.json"
python3 tools/pmc_build.py "$OUT/pmc_build.json" > "$OUT/pmc_build.txt" 2>&1
BFT_GPU_TRACE_IO=1 python3 tools/bench_bft_file.py 100 2> "$OUT/bft_file_100_trace.txt" | strip | tail -n 1 > "$OUT/bft_file_100.json"
python3 tools/bench_bft_file.py 10 2>&1 | strip | tail -n 1 > "$OUT/bft_file_10.json"
python3 tools/bench_color_lists.py cfg4 2>&1 | strip | grep "^{" > "$OUT/color_lists_cfg4.json"
python3 tools/bench_color_lists.py cfg2 100000000 2>&1 | strip | grep "^{" > "$OUT/color_lists_cfg2.json"
bash tools/profile_build_trace.sh "${1:-final}/build_config3" --reserve > /dev/null 2>&1
# one build as a timeline: kernels with start offsets and idle gaps (tools/build_timeline.py), and the host's own marks (BFT_GPU_TRACE_BUILD)
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/build_tl" -- python3 "$ROOT/tools/bench_insert.py" --reserve --sample 100000 > /dev/null 2>&1 )
python3 tools/build_timeline.py "$OUT/build_tl" > "$OUT/build_config3_timeline.txt" 2>&1
rm -rf "$OUT/build_tl"
BFT_GPU_TRACE_BUILD=1 python3 tools/bench_insert.py --reserve --sample 100000 2>&1 >/dev/null | grep "bft_gpu build" | tail -n 26 > "$OUT/build_config3_host_marks.txt"
BFT_GPU_TRACE_BUILD=1 python3 tools/bench_insert.py --reserve --k 31 --sample 100000 2>&1 >/dev/null | grep "bft_gpu build" | tail -n 26 > "$OUT/build_config3_k31_host_marks.txt"
python3 tools/bench_sequences.py 2>&1 | strip | grep "^{" > "$OUT/sequences.json"
python3 tools/bench_color_rows.py 2>&1 | strip | grep "^{" > "$OUT/color_rows.jsonl"
python3 tools/bench_color_rows.py cfg4 2>&1 | strip | grep "^{" >> "$OUT/color_rows.jsonl"
# config 5's kernels under the counters (branching through the k-mer hash, colour rows), and the microbenchmarks they are priced against
bash tools/pmc_collect_config5.sh > "$OUT/pmc_config5.log" 2>&1
cp gpurun_out/pmc/config5/pmc_k_branching.json "$OUT/pmc_config5_branching.json"; cp gpurun_out/pmc/config5/pmc_k_color_rows.json "$OUT/pmc_config5_color_rows.json"
cp gpurun_out/pmc/config5/kernel_stats.txt "$OUT/config5_kernel_stats.txt"
tools/microbench/gather 2 64 1024 8192 > "$OUT/microbench_gather.jsonl" 2>&1
tools/microbench/stream > "$OUT/microbench_stream.jsonl" 2>&1
[ -x tools/microbench/seg_sort ] && tools/microbench/seg_sort > "$OUT/microbench_seg_sort.jsonl" 2>&1
[ -x tools/microbench/kh_sort ] && tools/microbench/kh_sort > "$OUT/microbench_kh_sort.jsonl" 2>&1
[ -x tools/microbench/rs_sort ] && tools/microbench/rs_sort bench > "$OUT/microbench_rs_sort.jsonl" 2>&1
python3 tools/probe_fill.py 27 2>&1 | strip | grep "^{" > "$OUT/kmer_hash_build.json"
ls -la "$OUT"
