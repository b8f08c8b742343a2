#!/usr/bin/env python3
"""Config 5 (BASELINE.json): k=63, 2000-colour pan-genome (2000 x 20 kbp variants of one ancestor, ~4x10^7
(k-mer, genome) pairs), -query_branching + colour-set return on one MI355X.  Parity on a sample against ground
truth computed with torch from the per-genome k-mer tables."""
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
    ap.add_argument("--queries", type=int, default=10_000_000)
    ap.add_argument("--check", type=int, default=3000)
    ap.add_argument("--residencies", action="store_true", help="also time branching / presence under each query_wgs_per_cu setting")
    args = ap.parse_args()
    import torch
    from bloomfiltertrie_amd import BFT, synth as S
    k = args.k
    anc = S.random_genome(args.genome_len, 77)
    with BFT(k) as warm:  # loads the code objects and the library sort kernels once (~25 ms on the first call of a process; bench.py and
        w0 = S.distinct(S.kmers_of(anc, k))  # bench_insert.py do the same): not part of any figure
        warm.set_option("build_msd", 2)
        warm.insert_kmers(w0, 0)
        warm.insert_kmers(w0[::2], 1)
        warm.build()
        warm.query_presence(w0[:1000])
    t = BFT(k)
    gk = []
    t0 = time.perf_counter()
    for g in range(args.genomes):
        km = S.distinct(S.kmers_of(S.mutate(anc, args.snp_rate, 5000 + g), k))
        gk.append(km)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
    t.build()
    t_build = time.perf_counter() - t0
    info = t.info()
    union = S.distinct(np.concatenate(gk))
    assert info["kmers"] == len(union), (info["kmers"], len(union))
    # queries: 50% present, 50% SNP mutants
    rng = np.random.default_rng(1)
    nq = args.queries
    idx = rng.integers(0, len(union), nq)
    q = union[idx]
    half = nq // 2
    q[half:] = S.snp_mutants(q[half:], k, 9)
    q = np.ascontiguousarray(q[rng.permutation(nq)])
    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(q).to(dev)
    dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
    dcnt = torch.zeros(nq, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    lib = t._lib
    from bloomfiltertrie_amd import _lib as L
    for _ in range(2):
        L.check(lib.bft_gpu_query_branching_dev(t._h, dq.data_ptr(), nq, dbits.data_ptr(), None, stream))
    torch.cuda.synchronize()
    t.kernel_time(reset=True)
    for _ in range(5):
        L.check(lib.bft_gpu_query_branching_dev(t._h, dq.data_ptr(), nq, dbits.data_ptr(), None, stream))
    torch.cuda.synchronize()
    ms_b, n_b = t.kernel_time(reset=True)
    for _ in range(3):
        t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    ms_p, n_p = t.kernel_time(reset=True)
    per_res = {}
    if args.residencies:
        for r in (1, 2, 3):
            t.set_option("query_wgs_per_cu", r)
            for fn, name in ((lambda: L.check(lib.bft_gpu_query_branching_dev(t._h, dq.data_ptr(), nq, dbits.data_ptr(), None, stream)), "branching"),
                             (lambda: t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream), "presence")):
                fn()
                torch.cuda.synchronize()
                t.kernel_time(reset=True)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                ms_r, n_r = t.kernel_time(reset=True)
                per_res[f"{name}_ms_residency{r}"] = round(ms_r / n_r, 3)
        t.set_option("query_wgs_per_cu", 0)
    # colour rows, device resident: presence + ceil(G/8)-byte bitmap row per k-mer
    rowbytes = (args.genomes + 7) // 8
    nqc = min(nq, 4_000_000)
    drows = torch.zeros((nqc, rowbytes), dtype=torch.uint8, device=dev)
    dscr = torch.zeros(nqc, dtype=torch.int32, device=dev)
    for _ in range(2):
        L.check(lib.bft_gpu_query_color_rows_dev(t._h, dq.data_ptr(), nqc, dbits.data_ptr(), drows.data_ptr(), dscr.data_ptr(), stream))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        L.check(lib.bft_gpu_query_color_rows_dev(t._h, dq.data_ptr(), nqc, dbits.data_ptr(), drows.data_ptr(), dscr.data_ptr(), stream))
    torch.cuda.synchronize()
    t_rows_dev = (time.perf_counter() - t0) / 5
    t.kernel_time(reset=True)
    # colour sets through the host API on a slice (output-bound: ~1000 ids per present k-mer)
    ns = min(nq, 200_000)
    t0 = time.perf_counter()
    bits, off, ids = t.query_colors(q[:ns])
    t_col = time.perf_counter() - t0
    t0 = time.perf_counter()
    bits2, rows = t.query_color_rows(q[:ns])
    t_rows = time.perf_counter() - t0
    # ---- parity on a sample: ground truth from python sets ----
    nc = args.check
    truth = {}
    for g, km in enumerate(gk):
        for key in map(bytes, km):
            truth.setdefault(key, []).append(g)
    ok_col = True
    pres = S.from_bits(bits, ns)
    unp = np.unpackbits(rows[:nc], axis=1, bitorder="little")[:, :args.genomes]
    for i in range(nc):
        exp = truth.get(bytes(q[i]), [])
        ok_col &= pres[i] == bool(exp) and ids[int(off[i]):int(off[i + 1])].tolist() == exp and np.flatnonzero(unp[i]).tolist() == exp
    bbits, bcounts = t.query_branching(q[:nc], with_counts=True)
    codes = S.unpack_codes(q[:nc], k)
    ok_br = True
    for i in range(nc):
        c = codes[i]
        succ = sum(bytes(S.pack_codes(np.concatenate([c[1:], [x]])[None, :])[0]) in truth for x in range(4))
        pred = sum(bytes(S.pack_codes(np.concatenate([[x], c[:-1]])[None, :])[0]) in truth for x in range(4))
        ok_br &= int(bcounts[i]) == (succ << 4 | pred)
    out = {
        "workload": f"k={k}, {args.genomes} colours x {args.genome_len} nt ({args.snp_rate:.0%} SNPs), {nq} queries (50% present / 50% SNP mutants)",
        "pairs": info["pairs"], "distinct_kmers": info["kmers"], "colorsets": info["colorsets"],
        "kmer_gen_s": round(t_gen, 2), "insert_build_s": round(t_build, 3), "M_pairs_per_s": round(info["pairs"] / t_build / 1e6, 1),
        "build_breakdown_ms": {k_: round(v, 1) for k_, v in t.build_time().items()},
        "branching_M_kmers_per_s": round(nq / (ms_b / n_b) / 1e3, 1), "branching_ms": round(ms_b / n_b, 3),
        "branching_fraction": round(float(np.unpackbits(dbits.cpu().numpy(), bitorder='little')[:nq].mean()), 4),
        "presence_M_kmers_per_s": round(nq / (ms_p / n_p) / 1e3, 1),
        "color_rows_dev": {"queries": nqc, "row_bytes": rowbytes, "ms": round(t_rows_dev * 1e3, 3), "M_kmers_per_s": round(nqc / t_rows_dev / 1e6, 1),
                           "GB_per_s_written": round(nqc * rowbytes / t_rows_dev / 1e9, 1)},
        "colors_host_api": {"queries": ns, "ids_returned": int(len(ids)), "s": round(t_col, 3), "M_kmers_per_s": round(ns / t_col / 1e6, 3),
                            "rows_s": round(t_rows, 3), "rows_M_kmers_per_s": round(ns / t_rows / 1e6, 3)},
        "trie": {x: info[x] for x in ("nodes", "ccs", "child_nodes", "prefixes", "uc_rows", "root_ccs", "image_bytes")},
        **per_res,
        "parity": {"colors_sample": bool(ok_col), "branching_sample": bool(ok_br), "sample": nc},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
