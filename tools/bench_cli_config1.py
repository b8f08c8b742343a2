#!/usr/bin/env python3
"""Config 1 (BASELINE.json) through the C harness with the reference's command line: `bft_gpu build 27 kmers list out.bft`
on the 999 974 distinct 27-mers of a random 1 Mbp genome, then `bft_gpu load out.bft -query_kmers kmers list` with 10^6
ASCII k-mers (half present).  Wall times of the two processes, file size, and the printed count; the survey's probe of
the reference binary on the same input: build 5.78 s, query 1.07 s, 6.36 MB."""
import json
import os
import random
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bloomfiltertrie_amd import _lib  # noqa: E402

CLI = os.path.join(_lib.CSRC, "bft_gpu")
k = 27
random.seed(1)
genome = "".join(random.choice("ACGT") for _ in range(1_000_000))
seen, kmers = set(), []
for i in range(len(genome) - k + 1):
    s = genome[i:i + k]
    if s not in seen:
        seen.add(s)
        kmers.append(s)
rng = random.Random(2)
present = rng.sample(kmers, 500_000)
absent = []
while len(absent) < 500_000:
    s = "".join(rng.choice("ACGT") for _ in range(k))
    if s not in seen:
        absent.append(s)
queries = present + absent
rng.shuffle(queries)
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "genome.kmers"), "w").write("\n".join(kmers) + "\n")
    open(os.path.join(d, "list.txt"), "w").write(os.path.join(d, "genome.kmers") + "\n")
    open(os.path.join(d, "queries.txt"), "w").write("\n".join(queries) + "\n")
    open(os.path.join(d, "qlist.txt"), "w").write(os.path.join(d, "queries.txt") + "\n")
    subprocess.run([CLI, "--version"], capture_output=True)  # page the binary in
    t0 = time.perf_counter()
    r1 = subprocess.run([CLI, "build", str(k), "kmers", "list.txt", "out.bft"], capture_output=True, text=True, cwd=d)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    r2 = subprocess.run([CLI, "load", "out.bft", "-query_kmers", "kmers", "qlist.txt"], capture_output=True, text=True, cwd=d)
    t_query = time.perf_counter() - t0
    size = os.path.getsize(os.path.join(d, "out.bft"))
    csv_lines = open(os.path.join(d, "queries.csv"), "rb").read().count(b"\n") + 1
count = [x for x in r2.stdout.split("\n") if x.startswith("Nb k-mers present")]
print(json.dumps({"workload": "config 1: 999974 distinct 27-mers of a 1 Mbp genome, 10^6 ASCII queries (half present)", "kmers": len(kmers),
                  "build_process_s": round(t_build, 3), "load_and_query_process_s": round(t_query, 3), "bft_bytes": size, "csv_lines": csv_lines,
                  "stdout": count[0] if count else r2.stdout[-200:] + r2.stderr[-200:],
                  "reference_probe": {"build_s": 5.78, "load_and_query_s": 1.07, "bft_bytes": 6.36e6, "stdout": "Nb k-mers present = 500000"},
                  "rc": [r1.returncode, r2.returncode]}))
