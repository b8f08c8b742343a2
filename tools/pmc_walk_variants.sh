cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/pmc_live.py cfg4 125000000 gpurun_out/pmc_walk_q0.json kmer_hash=0 root_quartiles=0 > gpurun_out/pmc_walk_q0.log 2>&1
python tools/pmc_live.py cfg4 125000000 gpurun_out/pmc_walk_q1.json kmer_hash=0 root_quartiles=1 > gpurun_out/pmc_walk_q1.log 2>&1
python tools/pmc_live.py cfg4 125000000 gpurun_out/pmc_walk_q1_w1.json kmer_hash=0 root_quartiles=1 query_wgs_per_cu=1 > gpurun_out/pmc_walk_q1_w1.log 2>&1
tail -c 1500 gpurun_out/pmc_walk_q0.log; echo; tail -c 1500 gpurun_out/pmc_walk_q1.log; echo; tail -c 1500 gpurun_out/pmc_walk_q1_w1.log
