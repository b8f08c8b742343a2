#!/usr/bin/env python3
"""Algorithmic bytes per query (B(k) + 1/8 + S, S = trie bytes the REFERENCE algorithm dereferences, measured by the
oracle's counting mode) for the config-4 workload: the 100-genome index of bloomfiltertrie_amd/workloads.py and its query generator.
CPU only (the oracle is the checker; torch on the CPU generates the same seeded genomes as the GPU runs do -- the torch CPU and GPU
generators differ, so the index is statistically, not bit-wise, the one bench.py builds; S is a mean over 10^6 queries either way).
Writes profiles/r02_alg_bytes_config4.json, which bench.py reads for the roofline of the config-4 lines."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import synth as S, workloads as W  # noqa: E402
from oracle import oracle as O  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 27
genomes = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nq = 1_000_000
dev = torch.device("cpu")
pan = W.PanGenome(genomes, 2_000_000, 0.01, 4242, dev)
cnt = O.OracleBFT(k, count=True)
keys = []
t0 = time.time()
for g in range(genomes):
    packed = W.pack_windows(pan.genome(g), k)
    km = S.distinct(packed.numpy())
    cnt.insert_kmers(km, g)
    keys.append(W.unique_keys(W.keys_of(packed)))
    if g % 10 == 9:
        print(f"genome {g + 1}/{genomes} inserted, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
allk = W.union_of(keys)
gen = torch.Generator(device=dev)
gen.manual_seed(99)
dq, qk = W.presence_batch(allk, k, nq, gen)
bits, c = cnt.query_presence_count(dq.numpy())
truth = W.member(allk, qk).numpy()
assert (S.from_bits(np.asarray(bits), nq).astype(bool) == truth).all()
S_mean = c["bytes"] / nq
out = {"workload": f"config 4: k={k}, {genomes} genomes x 2 Mbp, 1% SNPs; {nq} queries of workloads.presence_batch (50% stored / 50% SNP mutants)",
       "total": round(S.kmer_bytes(k) + 0.125 + S_mean, 2), "kmer_in": S.kmer_bytes(k), "bit_out": 0.125, "trie_S": round(S_mean, 2),
       "ccs_scanned": round(c["ccs_scanned"] / nq, 2), "levels": round(c["levels"] / nq, 3), "distinct_kmers": int(allk.numel()),
       "oracle_build_s": round(time.time() - t0, 1)}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_alg_bytes_config4.json"), "w"), indent=1)
print(json.dumps(out))
