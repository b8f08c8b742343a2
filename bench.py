#!/usr/bin/env python3
"""bench.py -- batched k-mer presence throughput of the MI355X-native Bloom Filter Trie path.

Metric (BASELINE.json): M k-mers/sec queried (presenceNode / isKmerPresent), 1/2/4/8 MI355X.
Workload (configs[1], SURVEY.md 8d "Config 2"): 10 synthetic genomes (one 2 Mbp random ancestor, each genome
= ancestor with 1 % i.i.d. SNPs), BFT resident in HBM, 10^8 batched presence queries per GPU (50 % sampled from
the union of the genomes' k-mers, 50 % single-SNP mutants of present k-mers, interleaved at random).
k = 27: the reference rejects k = 31 (k must be a multiple of 9, src/main.c:61-63), so 27 is the
oracle-checkable stand-in the survey prescribes.

A "step" = one pass of the hot path (one bft_gpu_query_presence_dev launch) over the whole resident batch,
followed for N > 1 by the RCCL all_gather of the presence bitmaps.  Inputs are resident in HBM before the timed
region.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--genomes", type=int, default=10)
    ap.add_argument("--genome-len", type=int, default=2_000_000)
    ap.add_argument("--snp-rate", type=float, default=0.01)
    ap.add_argument("--queries", type=int, default=100_000_000, help="queries per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries timed on the CPU baseline (0 = auto)")
    ap.add_argument("--verify", type=int, default=1_000_000, help="queries of the batch checked against ground truth")
    ap.add_argument("--no-k31", action="store_true", help="skip the secondary k=31 measurement (extension beyond the reference)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the secondary host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the bitmap all_gather even with one rank (path check)")
    ap.add_argument("--replicate", choices=["rebuild", "broadcast"], default="rebuild",
                    help="how every rank gets the trie: each rank builds it (default, no collective outside the bitmap gather) "
                         "or rank 0 builds it and one RCCL broadcast replicates the image")
    return ap.parse_args()


def build_genome_kmers(args):
    from bloomfiltertrie_amd import synth as S
    anc = S.random_genome(args.genome_len, 1234)
    out = []
    for g in range(args.genomes):
        genome = S.mutate(anc, args.snp_rate, 1000 + g)
        out.append(S.distinct(S.kmers_of(genome, args.k)))
    return out


def make_queries_on_device(union_kmers, k, n, seed, device):
    """50 % sampled present k-mers, 50 % single-SNP mutants, interleaved at random; built on the GPU with torch."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    U = torch.from_numpy(union_kmers).to(device)
    nb = U.shape[1]
    out = torch.empty((n, nb), dtype=torch.uint8, device=device)
    chunk = 1 << 24
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        idx = torch.randint(0, U.shape[0], (m,), generator=g, device=device)
        q = U[idx]
        mut = torch.rand(m, generator=g, device=device) < 0.5
        pos = torch.randint(0, k, (m,), generator=g, device=device)
        delta = torch.randint(1, 4, (m,), generator=g, device=device).to(torch.uint8)
        byte = (pos // 4).long()
        sh = (2 * (pos % 4)).to(torch.uint8)
        rows = torch.arange(m, device=device)
        cur = q[rows, byte]
        nt = (cur >> sh) & 3
        new = (nt + delta) & 3
        newbyte = (cur & ~(torch.full_like(cur, 3) << sh)) | (new << sh)
        q[rows, byte] = torch.where(mut, newbyte, cur)
        out[a:a + m] = q
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from bloomfiltertrie_amd import BFT, synth as S

    # ---- build the trie (replicated on every rank) through the product insertion path ----
    t_build0 = time.time()
    gk = build_genome_kmers(args)
    t_gen = time.time() - t_build0
    with BFT(args.k, device=local_rank) as warm:  # loads the code objects and the hipCUB kernels once (not part of any figure)
        warm.insert_kmers(gk[0][:100000], 0)
        warm.build()
        warm.query_presence(gk[0][:1000])
    t0 = time.time()
    bft = None
    if rank == 0 or not (use_dist and args.replicate == "broadcast"):
        bft = BFT(args.k, device=local_rank)
        for gid, km in enumerate(gk):
            bft.add_genome(f"genome_{gid}")
            bft.insert_kmers(km, gid)
        bft.build()
    t_insert = time.time() - t0
    if use_dist and args.replicate == "broadcast":
        from bloomfiltertrie_amd.dist import replicate_image
        bft = replicate_image(bft, local_rank, src=0, always_copy=args.force_dist and world == 1)
    info = bft.info()
    union = S.distinct(np.concatenate(gk))
    assert info["kmers"] == len(union), (info["kmers"], len(union))

    # ---- the query batch, resident in HBM ----
    nq = args.queries
    dq = make_queries_on_device(union, args.k, nq, 99 + rank, device)
    # two result buffers: the bitmap gather of step i (RCCL, its own stream) overlaps the query kernel of step i+1
    nbuf = 2 if use_dist else 1
    bits_buf = [torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=device) for _ in range(nbuf)]
    gath_buf = [torch.empty(bits_buf[0].numel() * world, dtype=torch.uint8, device=device) for _ in range(nbuf)] if use_dist else None
    pending = [None] * nbuf
    stream = torch.cuda.current_stream().cuda_stream
    step_no = [0]

    def step():
        b = step_no[0] % nbuf
        step_no[0] += 1
        if pending[b] is not None:
            pending[b].wait()  # the gather that last read this buffer
            pending[b] = None
        bft.query_presence_dev(dq.data_ptr(), nq, bits_buf[b].data_ptr(), stream)
        if use_dist:
            pending[b] = dist.all_gather_into_tensor(gath_buf[b], bits_buf[b], async_op=True)

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    bft.kernel_time(reset=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = bft.kernel_time(reset=True)
    dbits = bits_buf[(step_no[0] - 1) % nbuf]  # the last step's answers
    if use_dist:
        gathered = gath_buf[(step_no[0] - 1) % nbuf]
        assert torch.equal(gathered[rank * dbits.numel():(rank + 1) * dbits.numel()], dbits)
        if nbuf > 1 and step_no[0] > 1:
            assert torch.equal(bits_buf[0], bits_buf[1])  # every step answers the same batch
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- correctness of the measured batch: ground truth on a slice + popcount property ----
    nv = min(args.verify, nq)
    host_q = dq[:nv].cpu().numpy()
    got = S.from_bits(dbits[: (nv + 7) // 8].cpu().numpy(), nv)
    truth = S.member(host_q, union)
    parity_ok = bool((got == truth).all())
    n_present = int(torch.from_numpy(np.unpackbits(dbits.cpu().numpy(), bitorder="little")[:nq]).sum())

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    value = nq * world * args.steps / elapsed / 1e6
    out = {
        "metric": "M k-mers/sec queried (presenceNode)",
        "value": round(value, 3),
        "unit": "M k-mers/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"k={args.k} (stand-in for k=31: reference requires k%9==0), {args.genomes}-genome BFT resident in HBM, "
                        f"{nq:.0e} batched presence queries per GPU (50% present / 50% SNP mutants)",
            "k": args.k, "genomes": args.genomes, "genome_len": args.genome_len, "snp_rate": args.snp_rate,
            "queries_per_gpu": nq, "distinct_kmers": info["kmers"], "pairs": info["pairs"],
            "trie": {x: info[x] for x in ("nodes", "ccs", "child_nodes", "prefixes", "uc_rows", "root_ccs", "image_bytes")},
            "parallelism": f"query-shard x{world}, trie replicated ({args.replicate if use_dist else 'single copy'})",
        },
        "parity_ok": parity_ok,
        "present_fraction": round(n_present / nq, 4),
        "build": {"kmer_gen_s": round(t_gen, 2), "insert_build_s": round(t_insert, 2),
                  "M_pairs_per_s": round(info["pairs"] / t_insert / 1e6, 3), **{k_: round(v, 1) for k_, v in bft.build_time().items()}},
    }

    # ---- secondary: the host-buffer entry point (H2D + kernel + D2H through bft_gpu_query_presence); never `value` ----
    if world == 1 and not args.no_pcie:
        nh = min(nq, 20_000_000)
        hq = np.ascontiguousarray(dq[:nh].cpu().numpy())
        bft.query_presence(hq[:1000])
        t0 = time.perf_counter()
        hb = bft.query_presence(hq)
        th = time.perf_counter() - t0
        out["pcie_inclusive"] = {"value": round(nh / th / 1e6, 3), "unit": "M k-mers/s", "queries": nh,
                                 "same_bits": bool((hb == dbits[: (nh + 7) // 8].cpu().numpy()).all()),
                                 "note": "pageable host buffers in and out, one call; for reference only"}
        del hq

    # ---- secondary: the same workload at the k the metric names (k=31), an extension the reference cannot run ----
    if not args.no_k31 and world == 1:
        try:
            k31 = 31
            anc31 = S.random_genome(args.genome_len, 1234)
            gk31 = [S.distinct(S.kmers_of(S.mutate(anc31, args.snp_rate, 1000 + g), k31)) for g in range(args.genomes)]
            b31 = BFT(k31, device=local_rank)
            for gid, km in enumerate(gk31):
                b31.insert_kmers(km, gid)
            b31.build()
            u31 = S.distinct(np.concatenate(gk31))
            q31 = make_queries_on_device(u31, k31, nq, 77, device)
            bits31 = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=device)
            b31.query_presence_dev(q31.data_ptr(), nq, bits31.data_ptr(), stream)
            torch.cuda.synchronize()
            b31.kernel_time(reset=True)
            for _ in range(args.steps):
                b31.query_presence_dev(q31.data_ptr(), nq, bits31.data_ptr(), stream)
            torch.cuda.synchronize()
            ms31, n31 = b31.kernel_time(reset=True)
            nv31 = min(args.verify, nq)
            ok31 = bool((S.from_bits(bits31[: (nv31 + 7) // 8].cpu().numpy(), nv31) == S.member(q31[:nv31].cpu().numpy(), u31)).all())
            out["k31_extension"] = {"value": round(nq / (ms31 / n31) / 1e3, 3), "unit": "M k-mers/s", "ms_per_launch": round(ms31 / n31, 4),
                                    "distinct_kmers": int(len(u31)), "parity_vs_ground_truth": ok31,
                                    "note": "k=31 is rejected by the reference (k % 9 != 0): no oracle exists; checked against set membership"}
            b31.close()
            del q31, bits31
        except Exception as e:  # the headline line must not depend on the extension
            out["k31_extension"] = {"error": str(e)}

    # ---- CPU baseline (oracle "port") + algorithmic bytes per query from its counting mode ----
    alg_bytes = None
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        cores = os.cpu_count() or 1
        ns = args.cpu_sample or min(nq, 1_500_000 * cores)
        sample = dq[:ns].cpu().numpy()
        if world == 1:  # the timed CPU baseline is an N=1 figure
            t0 = time.time()
            orc = O.OracleBFT(args.k)
            for gid, km in enumerate(gk):
                orc.insert_kmers(km, gid)
            orc.freeze()
            t_obuild = time.time() - t0
            t0 = time.time()
            obits = orc.query_presence(sample, threads=cores)
            t_q = time.time() - t0
            n1 = min(ns, 2_000_000)
            t0 = time.time()
            orc.query_presence(sample[:n1], threads=1)
            t_q1 = time.time() - t0
            gpu_bits = dbits[: (ns + 7) // 8].cpu().numpy()
            out["oracle_parity_ok"] = bool((obits == gpu_bits).all())
            out["cpu_baseline"] = {
                "value": round(ns / t_q / 1e6, 3), "unit": "M k-mers/s", "cores": cores, "kind": "port",
                "sample": f"first {ns} queries of the same batch, oracle isKmerPresent loop over {cores} threads sharing one trie "
                          f"(1 thread: {n1 / t_q1 / 1e6:.3f} M k-mers/s); oracle sequential build {t_obuild:.1f}s",
                "single_thread": round(n1 / t_q1 / 1e6, 3),
            }
        # algorithmic bytes per query (the roofline's numerator): the oracle's counting mode on a slice of the same batch
        cnt = O.OracleBFT(args.k, count=True)
        for gid, km in enumerate(gk):
            cnt.insert_kmers(km, gid)
        nc = min(ns, 1_000_000)
        cbits, c = cnt.query_presence_count(sample[:nc])
        if world > 1:
            out["oracle_parity_ok"] = bool((np.asarray(cbits)[: nc // 8] == dbits[: nc // 8].cpu().numpy()).all())
        S_mean = c["bytes"] / nc
        alg_bytes = S.kmer_bytes(args.k) + 1.0 / 8.0 + S_mean
        out["algorithmic_bytes_per_query"] = {"total": round(alg_bytes, 2), "kmer_in": S.kmer_bytes(args.k), "bit_out": 0.125,
                                              "trie_S": round(S_mean, 2), "ccs_scanned": round(c["ccs_scanned"] / nc, 2),
                                              "levels": round(c["levels"] / nc, 3)}
    if alg_bytes is None:
        alg_bytes = float(os.environ.get("BFT_ALG_BYTES_PER_QUERY", "0")) or None
    avg_ms = kern_ms / max(1, launches)
    if alg_bytes:
        achieved = alg_bytes * nq / (avg_ms * 1e-3) / 1e9
        traffic = None
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pj):
            try:
                traffic = json.load(open(pj)).get("k_query_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                           "kernel": "k_query", "avg_launch_ms": round(avg_ms, 4), "launches": int(launches),
                           "alg_bytes_per_launch": round(alg_bytes * nq)}
    if use_dist:
        dist.destroy_process_group()
    # RCCL writes its banner through C stdio: flush that first so that the JSON line is the last line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
