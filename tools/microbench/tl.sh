ROOT=$(pwd); OUT=$ROOT/gpurun_out/tl; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/build_tl" -- python3 "$ROOT/tools/bench_insert.py" --reserve --sample 100000 > /dev/null 2>&1 )
python3 tools/build_timeline.py "$OUT/build_tl" > "$OUT/build_config3_timeline.txt" 2>&1
rm -rf "$OUT/build_tl"
BFT_GPU_TRACE_BUILD=1 python3 tools/bench_insert.py --reserve --sample 100000 2>&1 >/dev/null | grep "bft_gpu build" | tail -n 26 > "$OUT/build_config3_host_marks.txt"
