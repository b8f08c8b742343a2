#!/usr/bin/env python3
"""Per-call cost of the host-buffer entry points at the reference's own batch size: src/file_io.c:726-730 reads the
query file in 4096-byte chunks (585 k-mers at k=27) and would call bft_gpu_query_presence once per chunk."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bloomfiltertrie_amd import BFT, synth as S  # noqa: E402

k = 27
km = S.distinct(S.kmers_of(S.random_genome(1_000_000, 1), k))
t = BFT(k)
t.insert_kmers(km, 0)
t.build()
out = {}
for timing in (1, 0):
    t.set_option("timing", timing)
    for n in (585, 4096, 65536, 1 << 20):
        q = np.ascontiguousarray(km[:n])
        for _ in range(20):
            t.query_presence(q)
        reps = 2000 if n < 100000 else 200
        t0 = time.perf_counter()
        for _ in range(reps):
            t.query_presence(q)
        dt = (time.perf_counter() - t0) / reps
        out[f"timing={timing},n={n}"] = {"us_per_call": round(dt * 1e6, 1), "M_kmers_per_s": round(n / dt / 1e6, 2)}
t.set_option("timing", 0)
for n in (1, 8, 585):  # what get_kmer / get_neighbors of <bft/bft.h> cost: presence + row + colour-set id per k-mer
    q = np.ascontiguousarray(km[:n])
    for _ in range(20):
        t.query_rows(q)
    t0 = time.perf_counter()
    for _ in range(2000):
        t.query_rows(q)
    dt = (time.perf_counter() - t0) / 2000
    out[f"query_rows,n={n}"] = {"us_per_call": round(dt * 1e6, 1)}
print(json.dumps(out))
