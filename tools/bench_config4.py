#!/usr/bin/env python3
"""Config 4 per-GPU share (BASELINE.json): k=27, 100-genome BFT (replicated on every GPU in the 8-GPU run), 1.25x10^8
presence queries per GPU.  Single-GPU measurement of that share; trie built through the device insert path."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_insert import pack_windows, keys_of  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=100)
    ap.add_argument("--queries", type=int, default=125_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--opt", action="append", default=[], help="name=value handed to bft_gpu_set_option after the build (e.g. kmer_hash=0)")
    ap.add_argument("--k", type=int, default=27, help="27 = the reference-compatible stand-in; 31 = the k the north star names (extension, ground truth only)")
    args = ap.parse_args()
    import torch
    from bloomfiltertrie_amd import BFT
    k, glen = args.k, 2_000_000
    assert k <= 31
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    anc = torch.randint(0, 4, (glen,), generator=g, device=dev, dtype=torch.uint8)
    t = BFT(k)
    keys = []
    for gid in range(args.genomes):
        m = torch.rand(glen, generator=g, device=dev) < 0.01
        delta = torch.randint(1, 4, (glen,), generator=g, device=dev, dtype=torch.uint8)
        packed = pack_windows(torch.where(m, (anc + delta) & 3, anc), k)
        t.insert_kmers_dev(packed.data_ptr(), packed.shape[0], gid)
        keys.append(torch.unique(keys_of(packed)))
        del packed
    t.build()
    for o in args.opt:
        name, v = o.split("=")
        t.set_option(name, int(v))
    info = t.info()
    allk = torch.unique(torch.cat(keys))
    del keys
    nq = args.queries
    idx = torch.randint(0, allk.numel(), (nq,), generator=g, device=dev)
    qk = allk[idx]
    # half of the queries: one SNP
    mut = torch.rand(nq, generator=g, device=dev) < 0.5
    pos = torch.randint(0, k, (nq,), generator=g, device=dev)
    delta = torch.randint(1, 4, (nq,), generator=g, device=dev)
    nt = (qk >> (2 * pos)) & 3
    qk = torch.where(mut, (qk & ~(torch.full_like(qk, 3) << (2 * pos))) | (((nt + delta) & 3) << (2 * pos)), qk)
    dq = qk.view(torch.uint8).reshape(-1, 8)[:, :(2 * k + 7) // 8].contiguous()
    dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    # ground truth for every query
    p = torch.searchsorted(allk, qk).clamp(max=allk.numel() - 1)
    truth = allk[p] == qk
    got = torch.from_numpy(np.unpackbits(dbits.cpu().numpy(), bitorder="little")[:nq].astype(bool)).to(dev)
    ok = bool((got == truth).all())
    t.kernel_time(reset=True)
    for _ in range(args.reps):
        t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    ms, n = t.kernel_time(reset=True)
    print(json.dumps({"options": args.opt, "workload": f"k={k}, {args.genomes}-genome BFT, {nq} queries (50% present / 50% SNP mutants)", "ms": round(ms / n, 3),
                      "G_kmers_per_s": round(nq / (ms / n) / 1e6, 2), "parity_all_queries": ok, "present_fraction": round(float(truth.float().mean()), 4), "tuned": t.build_time(),
                      "trie": {x: info[x] for x in ("kmers", "pairs", "colorsets", "nodes", "ccs", "child_nodes", "prefixes", "uc_rows", "root_ccs", "image_bytes")},
                      "footprint": t.footprint()}))


if __name__ == "__main__":
    main()
