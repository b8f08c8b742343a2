#!/usr/bin/env python3
"""bft_gpu_extract (the dump behind iterate_over_kmers / -extract_kmers) on the config-2 index: every stored
k-mer in the reference's packed layout + its colour-set id, host buffers out; the set is checked against the input."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bloomfiltertrie_amd import BFT, synth as S  # noqa: E402

k = 27
anc = S.random_genome(2_000_000, 1234)
gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 1000 + g), k)) for g in range(10)]
t = BFT(k)
for g, km in enumerate(gk):
    t.insert_kmers(km, g)
t.build()
t.extract()
t0 = time.perf_counter()
km, cs = t.extract()
dt = time.perf_counter() - t0
union = S.distinct(np.concatenate(gk))
ok = len(km) == len(union) and (np.sort(S.row_keys(km)) == np.sort(S.row_keys(union))).all()
print(json.dumps({"workload": "config-2 index, every stored k-mer + colour-set id to host buffers", "kmers": int(len(km)), "s": round(dt, 4),
                  "M_kmers_per_s": round(len(km) / dt / 1e6, 1), "same_set_as_inserted": bool(ok)}))
