ROOT=$(pwd); OUT=$ROOT/gpurun_out/tl3; rm -rf $OUT; mkdir -p $OUT
tools/microbench/rs_sort bench 2>&1 | grep -E "root-prefix|bad\": [1-9]"
tools/microbench/rs_sort quick 2>&1 | grep -E "bad\": [1-9]" | head -3
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tl" -- python3 "$ROOT/tools/bench_insert.py" --reserve --sample 100000 > /dev/null 2>&1 )
python3 tools/build_timeline.py "$OUT/tl" 2>&1 | grep -E "k_rs_pass<unsigned long, bft_rs::NoVal|k_bucket_sort|k_bucket_emit|k_cs_sig|# kernels" | cut -c1-190
rm -rf "$OUT/tl"
for i in 1 2; do python3 tools/bench_insert.py --reserve --stages --warm-pool 2>/dev/null | tail -n 1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg3', d['build_s'], d['value'], d['roofline']['gpu_ms_main_stream'], [(s['stage'][:12], s['ms']) for s in d['stages'][:3]])"; done
