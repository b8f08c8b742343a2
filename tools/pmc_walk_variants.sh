# counter passes of tools/pmc_live.py for a few option sets of the query path (config-4 share); results under gpurun_out/
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  name=$(echo "$v" | tr ' =' '__')
  python tools/pmc_live.py cfg4 125000000 gpurun_out/pmc_${name}.json $v > gpurun_out/pmc_${name}.log 2>&1
  python - <<PY
import json
d=json.load(open("gpurun_out/pmc_${name}.json"))
print("${v}", {k:d.get(k) for k in ("l2_misses_per_query","l2_requests_per_query","ea_read_requests_per_query","hbm_bytes_per_query","kernel_us_under_pmc_mean","error")})
PY
done
