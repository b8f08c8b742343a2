#!/usr/bin/env python3
"""Variants of the k-mer hash presence kernel on the config-4 index (100 genomes, 1.25x10^8 queries): k-mers per lane, grid multiplier,
occupancy of the table.  usage: probe_kh.py [k]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 27
dev = torch.device("cuda", 0)
pan = W.PanGenome(100, 2_000_000, 0.01, 4242, dev)
t = BFT(k)
keys, _ = W.build_index(t, pan, k)
allk = W.union_of(keys)
del keys
g = torch.Generator(device=dev)
g.manual_seed(99)
nq = 125_000_000
dq, qk = W.presence_batch(allk, k, nq, g)
truth = W.member(allk, qk)
dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def run(reps=5):
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    ok = bool((W.bits_to_bool(dbits, nq) == truth).all())
    t.kernel_time(reset=True)
    for _ in range(reps):
        t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    ms, n = t.kernel_time(reset=True)
    return ms / n, ok


for load in (50, 40, 65):
    t.set_option("kmer_hash_load", load)
    for mult in (1, 2, 4, 16):
        if True:
            t.set_option("query_grid_mult", mult)
            ms, ok = run()
            print(json.dumps({"k": k, "load": load, "grid_x": 4 * mult, "ms": round(ms, 3), "G_kmers_per_s": round(nq / ms / 1e6, 2), "ok": ok,
                              "kh_bytes": t.footprint()["kmer_hash"]}), flush=True)
