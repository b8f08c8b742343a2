#!/usr/bin/env python3
"""Why does the presence kernel take 2.96 ms in bench.py's main loop and 2.69 ms in tools/pmc_query.py on the same index and batch?
Replays the bench's sequence with pieces switched on and off.  usage: probe_bench_gap.py [warm] [msdwarm] [names] [dqfirst]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, synth as S, workloads as W  # noqa: E402

flags = set(sys.argv[1:])
k, nq = 27, 125_000_000
dev = torch.device("cuda", 0)
if "warm" in flags or "msdwarm" in flags:
    with BFT(k) as warm:
        wk = S.distinct(S.kmers_of(S.random_genome(120000, 5), k))
        if "msdwarm" in flags:
            warm.set_option("build_msd", 2)
        warm.insert_kmers(wk, 0)
        warm.build()
        warm.query_presence(wk[:1000])
pan = W.PanGenome(100, 2_000_000, 0.01, 4242, dev)
t = BFT(k)
if "names" in flags:
    for g in range(100):
        t.add_genome(f"genome_{g}")
keys, _ = W.build_index(t, pan, k)
allk = W.union_of(keys)
if "keepkeys" not in flags:
    del keys
g = torch.Generator(device=dev)
g.manual_seed(99)
dq, qk = W.presence_batch(allk, k, nq, g)
dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
res = {}
for label in ("first", "second"):
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    t.kernel_time(reset=True)
    for _ in range(10):
        t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    ms, n = t.kernel_time(reset=True)
    res[label] = round(ms / n, 4)
# drift: the same launches over ~1.5 s
series = []
for _ in range(25):
    t.kernel_time(reset=True)
    for _ in range(20):
        t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    ms, n = t.kernel_time(reset=True)
    series.append(round(ms / n, 3))
res["series_20_launches_each"] = series
# the table re-derived into a fresh allocation
t.set_option("kmer_hash", 0)
t.set_option("kmer_hash", 1)
t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
torch.cuda.synchronize()
t.kernel_time(reset=True)
for _ in range(10):
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
torch.cuda.synchronize()
ms, n = t.kernel_time(reset=True)
res["rederived"] = round(ms / n, 4)
print(json.dumps({"flags": sorted(flags), **res, "kh_ptr_mod_2MiB": None}))
