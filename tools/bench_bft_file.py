#!/usr/bin/env python3
"""write_BFT / load_BFT on the config-2 index: bft_gpu_write_bft (device image -> reference .bft layout on the
host) and bft_gpu_load_bft (parse + bulk rebuild on the GPU); the reloaded index answers a sample identically."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bloomfiltertrie_amd import BFT, synth as S  # noqa: E402

k = 27
ngen = int(sys.argv[1]) if len(sys.argv) > 1 else 10  # 10: config 2; 100: the config-3 / config-4 index
anc = S.random_genome(2_000_000, 1234)
gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 1000 + g), k)) for g in range(ngen)]
t = BFT(k)
for g, km in enumerate(gk):
    t.insert_kmers(km, g)
t.build()
shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
with tempfile.TemporaryDirectory(dir=shm) as d:
    path = os.path.join(d, "x.bft")
    t0 = time.perf_counter()
    t.write_bft(path)
    tw = time.perf_counter() - t0
    size = os.path.getsize(path)
    import hashlib
    sha = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    os.environ["BFT_GPU_IO_THREADS"] = "1"  # the same file from one thread: byte-identical
    p1 = os.path.join(d, "one.bft")
    t0 = time.perf_counter()
    t.write_bft(p1)
    tw1 = time.perf_counter() - t0
    same_bytes = hashlib.sha256(open(p1, "rb").read()).hexdigest()[:16] == sha
    os.unlink(p1)
    del os.environ["BFT_GPU_IO_THREADS"]
    t0 = time.perf_counter()
    u = BFT.load_bft(path)
    tl = time.perf_counter() - t0
union = S.distinct(np.concatenate(gk))
q = np.concatenate([union[::50], S.snp_mutants(union[::70], k, 3)])
a, ra = t.query_color_rows(q)
b, rb = u.query_color_rows(q)
info = t.info()
print(json.dumps({"workload": f"{ngen}-genome index, k = {k}", "kmers": info["kmers"], "pairs": info["pairs"], "file_bytes": size, "file_sha256_16": sha,
                  "write_s": round(tw, 3), "write_s_one_thread": round(tw1, 3), "same_bytes_from_one_thread": same_bytes, "load_s": round(tl, 3),
                  "io_threads": min(32, os.cpu_count() or 8), "MB_per_s_write": round(size / tw / 1e6, 1), "M_kmers_per_s_write": round(info["kmers"] / tw / 1e6, 2),
                  "M_kmers_per_s_load": round(info["kmers"] / tl / 1e6, 2), "same_answers_after_reload": bool((a == b).all() and (ra == rb).all())}))
