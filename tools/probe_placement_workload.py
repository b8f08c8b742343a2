#!/usr/bin/env python3
"""Workload of tools/probe_placement_pmc.sh: the 100-genome index, then the k-mer hash re-derived several times into fresh memory
(BFT_GPU_POOL_MAX_MB=0 in the environment: every derivation is a new hipMalloc; an unrelated block of a different size is taken and
released in between so that the driver hands out another range), three launches of the presence kernel on each.  The launches of one
process fall into both timing regimes; the profiler records counters and durations per dispatch.
usage: probe_placement_workload.py [derivations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W  # noqa: E402

n_der = int(sys.argv[1]) if len(sys.argv) > 1 else 6
k, nq = 27, 125_000_000
dev = torch.device("cuda", 0)
pan = W.PanGenome(100, 2_000_000, 0.01, 4242, dev)
t = BFT(k)
keys, _ = W.build_index(t, pan, k)
allk = W.union_of(keys)
del keys
g = torch.Generator(device=dev)
g.manual_seed(99)
dq, qk = W.presence_batch(allk, k, nq, g)
dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
held = []
for d in range(n_der):
    for _ in range(3):
        t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    t.set_option("kmer_hash", 0)
    held.append(torch.empty((97 + 61 * d) << 20, dtype=torch.uint8, device=dev))  # shifts where the next table lands
    t.set_option("kmer_hash", 1)
print("done", n_der)
