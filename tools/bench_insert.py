#!/usr/bin/env python3
"""Config 3 (BASELINE.json): insertKmers build path, 100 synthetic genomes (~2x10^8 (k-mer, genome) pairs) on one
MI355X.  Genomes are generated and packed on the GPU with torch (input plumbing), inserted genome by genome through
bft_gpu_insert_kmers_dev (ids ascending, as the reference requires) and built in bulk by bft_gpu_build.
Parity at this size is checked through size-independent properties: the number of distinct k-mers, the number of
distinct (k-mer, genome) pairs, and presence + colour sets of a random sample against per-genome sorted tables."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pack_windows(codes, k):
    """codes: uint8 tensor [G] on device -> packed k-mers [G-k+1, ceil(2k/8)] (reference layout)."""
    import torch
    n = codes.numel() - k + 1
    nb = (2 * k + 7) // 8
    win = codes.unfold(0, k, 1)  # [n, k] view
    pad = torch.zeros((n, nb * 4), dtype=torch.uint8, device=codes.device)
    pad[:, :k] = win
    q = pad.view(n, nb, 4)
    return (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).contiguous()


def keys_of(packed):
    import torch
    n, nb = packed.shape
    assert nb <= 8
    pad = torch.zeros((n, 8), dtype=torch.uint8, device=packed.device)
    pad[:, :nb] = packed
    return pad.view(torch.int64).reshape(n)


HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E spec peak
HBM_ACHIEVABLE_GBS = 6300.0  # ... and what a streaming kernel reaches


def stage_table(t, pairs_in):
    """bft_gpu_build_stages of the last build -> rows with GB/s against the achievable and the peak HBM rate.  A stage's bytes are what its
    algorithm reads + writes, from its own array sizes (csrc: the bft_stage() calls); 0 = a chain of small kernels / host round trips."""
    rows = []
    for name, ms, by in t.build_stages():
        r = {"stage": name, "ms": round(ms, 3)}
        if by > 0 and ms > 0:
            gbs = by / (ms * 1e-3) / 1e9
            r.update({"bytes": round(by), "bytes_per_pair": round(by / max(1, pairs_in), 2), "GB_s": round(gbs, 1),
                      "frac_of_achievable": round(gbs / HBM_ACHIEVABLE_GBS, 3), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 3)})
        rows.append(r)
    return rows


def cpu_insert_baseline(k, genome_len, snp_rate, genomes=8):
    """The oracle's insertKmers (oracle/bft_oracle.c: orc_insert_kmers, the restated container selection of src/insertNode.c:18-423) on the first
    `genomes` genomes of the same generator family, one thread: pairs/s."""
    from oracle import oracle as O
    from bloomfiltertrie_amd import synth as S
    anc = S.random_genome(genome_len, 4242)
    o = O.OracleBFT(k)
    n = 0
    t = 0.0
    for gid in range(genomes):
        km = S.kmers_of(S.mutate(anc, snp_rate, 100 + gid), k)
        t0 = time.perf_counter()
        o.insert_kmers(km, gid)
        t += time.perf_counter() - t0
        n += len(km)
    st = o.stats() if hasattr(o, "stats") else {}
    o.close()
    return {"value": round(n / t / 1e6, 3), "unit": "M pairs/s", "cores": 1, "kind": "port",
            "sample": f"the oracle's insertKmers on {genomes} genomes of {genome_len} nt ({n} pairs, ids ascending), one thread, {t:.1f} s",
            "oracle_trie": {k_: st[k_] for k_ in list(st)[:6]} if isinstance(st, dict) else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--genomes", type=int, default=100)
    ap.add_argument("--genome-len", type=int, default=2_000_000)
    ap.add_argument("--snp-rate", type=float, default=0.01)
    ap.add_argument("--sample", type=int, default=2_000_000)
    ap.add_argument("--reserve", action="store_true", help="reserve the insertion log for the total up front instead of letting it grow by doubling")
    ap.add_argument("--sync-inserts", action="store_true", help="bft_gpu_insert_kmers_dev (synchronised per call) instead of the stream-ordered bft_gpu_insert_kmers_dev_async")
    ap.add_argument("--add-genome", action="store_true", help="after the build, insert one more genome and time the incremental build (a merge)")
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to bft_gpu_set_option before the inserts (repeatable)")
    ap.add_argument("--warm-pool", action="store_true", help="build the same index once before the measured build, on a handle that is freed again: the library's cache of "
                    "released device blocks then holds what a build of this size needs (steady state of a process that builds repeatedly; a first build pays hipMalloc for "
                    "gigabyte blocks inside its stages -- 1.7 ms on one box, 33 ms on another)")
    ap.add_argument("--stages", action="store_true", help="per-stage GPU time and bytes of the build (bft_gpu_build_stages)")
    ap.add_argument("--cpu-baseline", type=int, default=0, help="also time the oracle's insertKmers on this many genomes (one thread)")
    args = ap.parse_args()
    print(json.dumps(measure(args)))


def measure(args):
    """args: the namespace of main() (bench.py builds one for its `insert` block)"""
    import torch
    from bloomfiltertrie_amd import BFT
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    anc = torch.randint(0, 4, (args.genome_len,), generator=g, device=dev, dtype=torch.uint8)
    with BFT(args.k) as warm:  # loads the code objects once (≈25 ms on the first call of a process;
        w = pack_windows(anc[:50000], args.k)  # bench.py does the same): not part of any figure
        warm.set_option("build_msd", 2)  # (the root-prefix bucket kernels too: they otherwise load on the first large build)
        warm.insert_kmers_dev(w.data_ptr(), w.shape[0], 0)
        warm.build()
        warm.insert_kmers_dev(w.data_ptr(), w.shape[0] // 2, 1)  # (and the merge kernels: a second build on the same handle)
        warm.build()
        warm.query_presence(w[:1000].cpu().numpy())
        del w
    t = BFT(args.k)
    if args.stages:
        t.set_option("build_stages", 1)
    for o in args.opt:
        name, val = o.split("=")
        t.set_option(name, int(val))
    t_ins = 0.0
    if args.reserve:  # the total is known up front (genomes x windows), as with the count line of a kmers_comp file
        t0 = time.perf_counter()
        t.set_option("reserve_pairs", args.genomes * (args.genome_len - args.k + 1))
        t_ins += time.perf_counter() - t0
    npairs_in = 0
    per_genome_keys = []
    batches = []  # every batch is generated (and its ground truth taken) before the clock starts: the timed region is insert calls + build
    for gid in range(args.genomes):
        m = torch.rand(args.genome_len, generator=g, device=dev) < args.snp_rate
        delta = torch.randint(1, 4, (args.genome_len,), generator=g, device=dev, dtype=torch.uint8)
        genome = torch.where(m, (anc + delta) & 3, anc)
        packed = pack_windows(genome, args.k)
        npairs_in += packed.shape[0]
        per_genome_keys.append(torch.unique(keys_of(packed)))  # sorted
        batches.append(packed)
    stream = torch.cuda.current_stream().cuda_stream
    first_build_s = None
    if getattr(args, "warm_pool", False):
        with BFT(args.k) as w2:
            for o in args.opt:
                name, val = o.split("=")
                w2.set_option(name, int(val))
            w2.set_option("reserve_pairs", args.genomes * (args.genome_len - args.k + 1))
            for gid, packed in enumerate(batches):
                w2.insert_kmers_dev_async(packed.data_ptr(), packed.shape[0], gid, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            w2.build()
            first_build_s = time.perf_counter() - t0
        if args.reserve:
            t.set_option("reserve_pairs", args.genomes * (args.genome_len - args.k + 1))  # (the log again: the warm-up handle's went back to the cache)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for gid, packed in enumerate(batches):
        if args.sync_inserts:
            t.insert_kmers_dev(packed.data_ptr(), packed.shape[0], gid)  # synchronised per call
        else:
            t.insert_kmers_dev_async(packed.data_ptr(), packed.shape[0], gid, stream)  # stream-ordered: the build waits for the stream
    t_ins += time.perf_counter() - t0
    t0 = time.perf_counter()
    t.build()
    t_build = time.perf_counter() - t0
    breakdown = {k_: round(v, 1) for k_, v in t.build_time().items()}  # of THIS build (a later build of the handle overwrites the library's record)
    stages = stage_table(t, npairs_in) if args.stages else None
    t.set_option("build_stages", 0)
    info_built = t.info()
    del batches
    # one more genome onto the finished index (-add_genomes): its run is sorted and merged into the index, not everything re-sorted
    add = None
    if args.add_genome:
        m = torch.rand(args.genome_len, generator=g, device=dev) < args.snp_rate
        delta = torch.randint(1, 4, (args.genome_len,), generator=g, device=dev, dtype=torch.uint8)
        extra = pack_windows(torch.where(m, (anc + delta) & 3, anc), args.k)
        before = t.info()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.insert_kmers_dev_async(extra.data_ptr(), extra.shape[0], args.genomes, stream)
        t.build()
        t_add = time.perf_counter() - t0
        after = t.info()
        ek = torch.unique(keys_of(extra))
        per_genome_keys.append(ek)
        add = {"pairs": int(extra.shape[0]), "ms": round(t_add * 1e3, 2), "new_kmers": after["kmers"] - before["kmers"], "new_colorsets": after["colorsets"] - before["colorsets"],
               "build_breakdown_ms": {k_: round(v, 1) for k_, v in t.build_time().items() if k_.endswith("_ms") and v}}
        args.genomes += 1
        del extra
    info = t.info()
    # ---- properties ----
    allk = torch.unique(torch.cat(per_genome_keys))
    pairs = int(sum(int(x.numel()) for x in per_genome_keys))
    ok_counts = info["kmers"] == int(allk.numel()) and info["pairs"] == pairs
    # sample: half present, half random
    ns = args.sample
    idx = torch.randint(0, allk.numel(), (ns // 2,), generator=g, device=dev)
    qk = torch.cat([allk[idx], torch.randint(0, 1 << (2 * args.k), (ns - ns // 2,), generator=g, device=dev, dtype=torch.int64)])
    nb = (2 * args.k + 7) // 8
    q = qk.view(torch.uint8).reshape(-1, 8)[:, :nb].contiguous().cpu().numpy()
    bits, rows = t.query_color_rows(q)
    got = np.unpackbits(rows, axis=1, bitorder="little")[:, :args.genomes].astype(bool)
    exp = np.zeros_like(got)
    for gid, gkeys in enumerate(per_genome_keys):
        pos = torch.searchsorted(gkeys, qk).clamp(max=gkeys.numel() - 1)
        exp[:, gid] = (gkeys[pos] == qk).cpu().numpy()
    ok_colors = bool((got == exp).all())
    ok_presence = bool((np.unpackbits(bits, bitorder="little")[:ns].astype(bool) == exp.any(axis=1)).all())
    out = {
        "metric": "M (k-mer, genome) pairs/sec inserted (insertKmers bulk build)",
        "workload": f"k={args.k}, {args.genomes - (1 if add else 0)} genomes x {args.genome_len} nt, {args.snp_rate:.0%} SNPs, ids ascending"
                    + (" (+ one more genome added afterwards: add_one_genome; parity is checked on the index with it)" if add else ""),
        "reserved": args.reserve, "warmed_up": True, "inserts": "synchronised per call" if args.sync_inserts else "stream-ordered (bft_gpu_insert_kmers_dev_async)", "pairs_in": npairs_in, "pairs_distinct": info_built["pairs"], "distinct_kmers": info_built["kmers"], "colorsets": info_built["colorsets"],
        "insert_s": round(t_ins, 4), "build_s": round(t_build, 4), "first_build_of_the_process_s": None if first_build_s is None else round(first_build_s, 4),
        "value": round(npairs_in / (t_ins + t_build) / 1e6, 2), "unit": "M pairs/s",
        "build_breakdown_ms": breakdown,
        "trie": {x: info_built[x] for x in ("nodes", "ccs", "child_nodes", "prefixes", "uc_rows", "root_ccs", "image_bytes")},
        "add_one_genome": add,
        "parity": {"counts": bool(ok_counts), "presence_sample": ok_presence, "colors_sample": ok_colors, "sample": ns},
    }
    if stages is not None:
        main_ms = sum(r["ms"] for r in stages if not r["stage"].startswith("+"))
        by = sum(r.get("bytes", 0) for r in stages if not r["stage"].startswith("+"))
        out["stages"] = stages
        out["roofline"] = {"bound": "hbm", "achieved": round(by / (main_ms * 1e-3) / 1e9, 1) if main_ms else None, "peak": HBM_PEAK_GBS, "achievable": HBM_ACHIEVABLE_GBS,
                           "unit": "GB/s", "frac": round(by / (main_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if main_ms else None, "traffic": None,
                           "gpu_ms_main_stream": round(main_ms, 3), "algorithmic_bytes": round(by), "bytes_per_pair": round(by / max(1, npairs_in), 1),
                           "note": "whole build: the algorithmic bytes of the streaming stages (the others -- chains of small kernels, host round trips -- count as time with no bytes) "
                                   "over the GPU time of the build's main stream; per stage under `stages` ('+': the k-mer hash build on the side stream, from the build's start)"}
    if args.cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_insert_baseline(args.k, args.genome_len, args.snp_rate, args.cpu_baseline)
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    t.close()
    return out


if __name__ == "__main__":
    main()
