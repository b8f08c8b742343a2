#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes of tools/pmc_collect.sh: build the config-2 / config-4 index (workloads.py), then
`reps` launches of the presence kernel on the resident batch.  Options after the workload name: name=value pairs handed to
bft_gpu_set_option (e.g. kmer_hash=0 root_direct=0).
"sweepK" = the index of tools/bench_k_sweep.py at k = K (e.g. sweep63: two-word rows).
usage: pmc_query.py <cfg2|cfg4|cfg4k31|sweepK> <queries> <reps> [option=value ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
opts = [a.split("=") for a in sys.argv[4:]]
dev = torch.device("cuda", 0)
if wl.startswith("sweep"):
    import numpy as np  # noqa: E402
    from bloomfiltertrie_amd import synth as S  # noqa: E402
    from bloomfiltertrie_amd.workloads import make_queries_on_device  # noqa: E402
    k = int(wl[5:])
    anc = S.random_genome(2_000_000, 1234)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 1000 + g), k)) for g in range(10)]
    t = BFT(k)
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
    t.build()
    for name, v in opts:
        t.set_option(name, int(v))
    dq = make_queries_on_device(S.distinct(np.concatenate(gk)), k, nq, 5, dev)
else:
    k = 31 if wl.endswith("k31") else 27
    genomes = 10 if wl.startswith("cfg2") else 100
    pan = W.PanGenome(genomes, 2_000_000, 0.01, 4242, dev)
    t = BFT(k)
    keys, _ = W.build_index(t, pan, k)
    allk = W.union_of(keys)
    del keys
    for name, v in opts:
        t.set_option(name, int(v))
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    dq, qk = W.presence_batch(allk, k, nq, g)
dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
torch.cuda.synchronize()
print("done", wl, nq, t.info()["image_bytes"], t.build_time())
