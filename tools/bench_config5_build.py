#!/usr/bin/env python3
"""Config 5's insertion path on its own (BASELINE configs[4]: k = 63, 2000 colours x 20 kbp): 2000 host calls of insertKmers
(include/insertNode.h:26) with ~20000 k-mers each, then bft_gpu_build -- the calls and the build timed apart, the build stage by stage
(bft_gpu_build_stages), once on a fresh process state and once with the library's block cache warm.
usage: bench_config5_build.py [--k 63] [--genomes 2000] [--opt name=value ...]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=63)
    ap.add_argument("--genomes", type=int, default=2000)
    ap.add_argument("--genome-len", type=int, default=20000)
    ap.add_argument("--snp-rate", type=float, default=0.01)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    from bloomfiltertrie_amd import BFT, synth as S
    k = args.k
    anc = S.random_genome(args.genome_len, 77)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, args.snp_rate, 5000 + g), k)) for g in range(args.genomes)]
    pairs = int(sum(len(x) for x in gk))
    with BFT(k) as warm:  # code objects
        warm.set_option("build_msd", 2)
        warm.insert_kmers(gk[0], 0)
        warm.insert_kmers(gk[1], 1)
        warm.build()
    runs = []
    for rep in range(2):
        t = BFT(k)
        t.set_option("build_stages", 1)
        for o in args.opt:
            name, val = o.split("=")
            t.set_option(name, int(val))
        t0 = time.perf_counter()
        for g, km in enumerate(gk):
            t.insert_kmers(km, g)
        t_ins = time.perf_counter() - t0
        t0 = time.perf_counter()
        t.build()
        t_build = time.perf_counter() - t0
        info = t.info()
        st = [{"stage": n, "ms": round(ms, 3)} for n, ms, _ in t.build_stages()]
        bt = t.build_time()
        runs.append({"pool": "cold" if rep == 0 else "warm", "insert_calls_s": round(t_ins, 4), "us_per_call": round(t_ins / args.genomes * 1e6, 1), "build_s": round(t_build, 4),
                     "insert_build_s": round(t_ins + t_build, 4), "M_pairs_per_s": round(pairs / (t_ins + t_build) / 1e6, 1),
                     "gpu_ms_main_stream": round(sum(r["ms"] for r in st if not r["stage"].startswith("+")), 2), "stages": st,
                     "breakdown": {x: round(bt[x], 2) for x in ("gpu_sort_dedupe_ms", "color_intern_ms", "assemble_ms", "derive_ms", "kmer_hash_fill_ms", "sort_max_bucket", "process_hipmalloc_ms")}})
        if rep == 1:
            union = S.distinct(np.concatenate(gk))
            ok = info["kmers"] == len(union) and info["pairs"] == pairs
            rng = np.random.default_rng(3)
            q = np.ascontiguousarray(np.concatenate([union[rng.integers(0, len(union), 20000)], S.snp_mutants(union[rng.integers(0, len(union), 20000)], k, 4)]))
            bits = t.query_presence(q)
            ok = ok and bool((S.from_bits(bits, len(q)) == S.member(q, union)).all())
        t.close()
    print(json.dumps({"workload": f"k={k}, {args.genomes} colours x {args.genome_len} nt, {pairs} pairs, host insert calls of ~{pairs // args.genomes} k-mers", "kmers": info["kmers"],
                      "colorsets": info["colorsets"], "nodes": info["nodes"], "runs": runs, "counts_and_presence_sample_ok": ok}))


if __name__ == "__main__":
    main()
