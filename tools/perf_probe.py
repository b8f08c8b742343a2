#!/usr/bin/env python3
"""Development probe (GPU box): time k_query under the tuning knobs on several workloads, checking results each time.

usage: python tools/perf_probe.py [--queries N] [--workloads cfg2,deep,k63]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workload(name):
    from bloomfiltertrie_amd import synth as S
    if name == "cfg2":
        k = 27
        anc = S.random_genome(2_000_000, 1234)
        gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 1000 + g), k)) for g in range(10)]
    elif name == "small":  # the whole table fits the 4 MiB L2 of one XCD
        k = 27
        gk = [S.distinct(S.kmers_of(S.random_genome(300_000, 5), k))]
    elif name == "mid":  # fits the 256 MiB Infinity Cache, not L2
        k = 27
        gk = [S.distinct(S.kmers_of(S.random_genome(3_000_000, 6), k))]
    elif name == "deep":  # low-entropy: three trie levels are exercised
        k = 27
        gk = [S.low_entropy_kmers(6_000_000, k, 600, seed=7, levels=1)]
    elif name == "deep2":
        k = 27
        gk = [S.low_entropy_kmers(6_000_000, k, 40, seed=8, levels=2)]
    elif name == "k63":
        k = 63
        anc = S.random_genome(1_000_000, 4321)
        gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 2000 + g), k)) for g in range(4)]
    else:
        raise SystemExit(name)
    return k, gk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=50_000_000)
    ap.add_argument("--workloads", default="cfg2,deep,deep2,k63")
    ap.add_argument("--mults", default="1,2,4")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--wgs", default="0", help="comma list of query_wgs_per_cu settings (0 = automatic)")
    ap.add_argument("--probes", default="0", help="comma list of query_probe settings (0 = automatic, 4, 8)")
    ap.add_argument("--flat-mins", default="3584", help="comma list of flat_min settings to time (65536 = flat form off)")
    ap.add_argument("--bucket-bits", type=int, default=0, help="experiment: stable-sort the query batch by the top bits of its rotated root prefix (n2..) before timing")
    ap.add_argument("--node-hash", default="1", help="comma list of 0/1: node prefix hash below the root")
    ap.add_argument("--stops", action="store_true", help="time the walk truncated after each stage (results wrong)")
    args = ap.parse_args()
    import torch
    from bloomfiltertrie_amd import BFT, synth as S
    sys.path.insert(0, ROOT)
    from bloomfiltertrie_amd.workloads import make_queries_on_device
    dev = torch.device("cuda", 0)
    for wl in args.workloads.split(","):
        k, gk = workload(wl)
        t = BFT(k)
        t0 = time.time()
        for g, km in enumerate(gk):
            t.insert_kmers(km, g)
        t.build()
        info = t.info()
        union = S.distinct(np.concatenate(gk)) if len(gk) > 1 else gk[0]
        nq = args.queries
        dq = make_queries_on_device(union, k, nq, 5, dev)
        if args.bucket_bits:
            nn = (args.bucket_bits + 1) // 2
            key = torch.zeros(nq, dtype=torch.int32, device=dev)
            for j in range(1, 1 + nn):  # n2, n3, ...: nucleotide j (0-based) sits in byte j//4, bits 2*(j%4)
                key = (key << 2) | ((dq[:, j // 4].to(torch.int32) >> (2 * (j % 4))) & 3)
            order = torch.sort(key, stable=True).indices
            dq = dq[order].contiguous()
            del key, order
        dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        nv = 500_000
        truth = S.member(dq[:nv].cpu().numpy(), union)
        print(json.dumps({"workload": wl, "k": k, "build_s": round(time.time() - t0, 2), **{x: info[x] for x in ("kmers", "nodes", "ccs", "child_nodes", "uc_rows", "root_ccs", "image_bytes")}}), flush=True)
        for nh, fm, mult, wg, pr in [(n_, f, m, w, p) for n_ in [int(x) for x in args.node_hash.split(",")] for f in [int(x) for x in args.flat_mins.split(",")]
                                          for m in [int(x) for x in args.mults.split(",")] for w in [int(x) for x in args.wgs.split(",")]
                                          for p in [int(x) for x in args.probes.split(",")]]:
            if True:
                t.set_option("node_hash", nh)
                t.set_option("query_probe", pr)
                t.set_option("flat_min", fm)
                t.set_option("query_wgs_per_cu", wg)
                t.set_option("query_grid_mult", mult)
                dbits.zero_()
                t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
                torch.cuda.synchronize()
                ok = bool((S.from_bits(dbits[: (nv + 7) // 8].cpu().numpy(), nv) == truth).all())
                t.kernel_time(reset=True)
                for _ in range(args.reps):
                    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
                torch.cuda.synchronize()
                ms, n = t.kernel_time(reset=True)
                print(json.dumps({"workload": wl, "node_hash": nh, "flat_min": fm, "image_bytes": t.info()["image_bytes"], "wgs_per_cu": wg, "probe": pr, "in_use": [int(t.build_time()["query_wgs_per_cu"]), int(t.build_time()["query_probe_rows"])], "grid_mult": mult, "ms": round(ms / n, 3), "Gq_s": round(nq / (ms / n) / 1e6, 2), "ok": ok}), flush=True)
        if args.stops:  # needs BFT_GPU_LIB=bloomfiltertrie_amd/csrc/libbft_gpu_probe.so (make -C bloomfiltertrie_amd/csrc probe)
            t.set_option("query_grid_mult", 1)
            for stop in (1, 2, 3, 4, 0):
                t.set_option("debug_stop", stop)
                t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
                torch.cuda.synchronize()
                t.kernel_time(reset=True)
                for _ in range(args.reps):
                    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
                torch.cuda.synchronize()
                ms, n = t.kernel_time(reset=True)
                print(json.dumps({"workload": wl, "debug_stop": stop, "ms": round(ms / n, 3)}), flush=True)
        t.close()
        del dq, dbits
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
