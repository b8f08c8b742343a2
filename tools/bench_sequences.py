#!/usr/bin/env python3
"""Sequence queries (query_sequence of include/bft.h:127) on the config-2 index: reads of 150 nt sampled from the
genomes (1 % errors) plus random reads, threshold 0.8; a sample of the rows is checked against the oracle's restatement."""
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
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genomes", type=int, default=10)
    ap.add_argument("--check", type=int, default=300)
    args = ap.parse_args()
    from bloomfiltertrie_amd import BFT, synth as S
    from oracle import oracle as O
    k, glen = 27, 2_000_000
    anc = S.random_genome(glen, 1234)
    genomes = [S.mutate(anc, 0.01, 1000 + g) for g in range(args.genomes)]
    t = BFT(k)
    o = O.OracleBFT(k)
    for g, gen in enumerate(genomes):
        km = S.distinct(S.kmers_of(gen, k))
        t.insert_kmers(km, g)
        if args.check:
            o.insert_kmers(km, g)
    t.build()
    rng = np.random.default_rng(5)
    n, rl = args.reads, args.read_len
    src = rng.integers(0, args.genomes, n)
    start = rng.integers(0, glen - rl, n)
    G = np.stack(genomes)
    idx = start[:, None] + np.arange(rl)[None, :]
    reads = G[src[:, None], idx]
    err = rng.random((n, rl)) < 0.01
    reads = np.where(err, (reads + rng.integers(1, 4, (n, rl))) & 3, reads).astype(np.uint8)
    reads[n - n // 10:] = rng.integers(0, 4, (n // 10, rl), dtype=np.uint8)  # 10 % random reads
    ascii_reads = np.frombuffer(b"ACGT", dtype=np.uint8)[reads]
    seqs = [bytes(r) for r in ascii_reads]
    t.query_sequences(seqs[:1000], 0.8)  # warm-up
    # the C call alone (bft_gpu_query_sequences: host ASCII blob + offsets in, genome-bit rows out); the Python mirror's list
    # building is not part of the figure
    import ctypes as C
    from bloomfiltertrie_amd import _lib
    blob = ascii_reads.tobytes() + b"\0"
    off = (np.arange(n + 1, dtype=np.uint64) * rl)
    rowbytes = (args.genomes + 7) // 8
    out = {}
    for canonical in (False, True):
        rows = np.zeros((n, rowbytes), dtype=np.uint8)
        dts = []
        for _ in range(3):  # the first call of a size also pays for its device buffers (cached afterwards)
            t0 = time.perf_counter()
            _lib.check(t._lib.bft_gpu_query_sequences(t._h, blob, off.ctypes.data, n, C.c_double(0.8), int(canonical), rows.ctypes.data))
            dts.append(time.perf_counter() - t0)
        dt = min(dts)
        nk = n * (rl - k + 1)
        unp = np.unpackbits(rows, axis=1, bitorder="little")[:, :args.genomes]
        ok = True
        for i in list(range(0, args.check // 2)) + list(range(n - args.check // 2, n)):
            ok &= np.flatnonzero(unp[i]).tolist() == o.query_sequence(seqs[i].decode(), 0.8, canonical, args.genomes)
        # the same batch resident in HBM (bft_gpu_query_sequences_dev): what the kernels cost without the PCIe trips
        import torch
        dev = torch.device("cuda", 0)
        d_blob = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
        d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
        d_rows = torch.zeros((n, rowbytes), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        t.query_sequences_dev(d_blob.data_ptr(), d_off.data_ptr(), n, n * rl, 0.8, d_rows.data_ptr(), canonical, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            t.query_sequences_dev(d_blob.data_ptr(), d_off.data_ptr(), n, n * rl, 0.8, d_rows.data_ptr(), canonical, st)
        e1.record()
        torch.cuda.synchronize()
        dev_ms = e0.elapsed_time(e1) / 5
        dev_same = bool((d_rows.cpu().numpy() == rows).all())
        out["canonical" if canonical else "as_is"] = {"device_resident": {"ms": round(dev_ms, 3), "M_reads_per_s": round(n / dev_ms / 1e3, 1), "M_kmers_per_s": round(nk / dev_ms / 1e3, 1),
                                                                          "rows_equal_host_call": dev_same},
                                                      "s": round(dt, 4), "M_reads_per_s": round(n / dt / 1e6, 3), "M_kmers_per_s": round(nk / dt / 1e6, 1), "calls_s": [round(x, 4) for x in dts],
                                                      "reads_with_a_genome": int(unp.any(axis=1).sum()), "oracle_parity_on_sample": bool(ok)}
    print(json.dumps({"workload": f"k={k}, {args.genomes}-genome index, {n} reads x {rl} nt (1% errors, 10% random), threshold 0.8; top-level figures: host buffers in and out, device_resident: bft_gpu_query_sequences_dev", **out}))


if __name__ == "__main__":
    main()
