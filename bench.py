#!/usr/bin/env python3
"""bench.py -- batched k-mer presence throughput of the MI355X-native Bloom Filter Trie path.

Metric (BASELINE.json): M k-mers/sec queried (presenceNode / isKmerPresent), 1/2/4/8 MI355X; HBM GB/s vs peak.

ONE workload at every N (so that N = 1 is the point an 8-GPU curve starts from): BASELINE configs[3] / the north star's index --
100 synthetic genomes (one 2 Mbp random ancestor, 1 % i.i.d. SNPs each; 4.5x10^7 distinct k-mers, 2x10^8 (k-mer, genome) pairs)
resident in every GPU's HBM, 1.25x10^8 = 10^9 / 8 batched presence queries per GPU (50 % stored k-mers, 50 % single-SNP mutants),
every answer of every shard checked against ground truth; for N > 1 the image is built on rank 0 and replicated by one RCCL broadcast
(or rebuilt per rank: --replicate rebuild) and the presence bitmaps are all_gathered over xGMI, the gather of step i overlapping the
kernel of step i+1.  k = 27: the reference rejects k = 31 (k must be a multiple of 9, src/main.c:61-63), so 27 is the
oracle-checkable stand-in the survey prescribes; the same index at k = 31 is measured beside it (`k31`, ground truth only).
Secondary blocks at N = 1 (never `value`): configs[1] (10-genome index, 10^8 queries), the container walk without the k-mer hash,
sequence queries, host buffers (PCIe), the footprint of the image against the .bft file, the CPU baseline.

A "step" = one pass of the hot path (one bft_gpu_query_presence_dev call) over the whole resident batch, followed for N > 1 by the
RCCL all_gather of the presence bitmaps.  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
`python bench.py --gpus N` without torchrun starts the N ranks itself (fresh child processes, before this process touches a GPU).

roofline: `traffic` = HBM bytes per launch from hardware counters collected IN THIS RUN (tools/pmc_live.py: rocprofv3 --pmc passes in
child processes after the timed region -- same library, same workload generator and seed as the timed batch --: FETCH_SIZE + WRITE_SIZE as
counted, `traffic_counted`, plus a calibrated correction for the coalesced query stream the fetch counter tallies at half, `stream_correction`),
`achieved` = traffic / mean kernel time (HIP events on the launch stream), `frac` = achieved / 8 TB/s -- an estimate resting on that
calibration, not a bound.  Without a profiler the figures come from profiles/r05/pmc_query.json when its source hash matches the library's,
else `frac` is null and `pmc_stale` true.  `gather` = L2 misses per second against the measured random-gather ceiling of the chip -- the
limit that binds this kernel.  The container walk ("walk_hash": plain root groups in hashed form, special prefixes
through the containers) gets the same block of its own under `container_walk.roofline`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak (6.3 TB/s achievable)
# Random gathers per second beyond the L2: the request rate of the fabric.  59.4 G/s with the table inside the 256 MiB Infinity Cache, 56.0 G/s
# on a 1 GiB table, 49 on an 8 GiB one (independent loads; whole lines fetched by quads of lanes the way k_query_kh does: 54.7 at 1 GiB) --
# profiles/r04/microbench_gather.jsonl.  The 0.67 GB k-mer hash of the config-4 share lies partly inside the Infinity Cache: the kernel is priced
# against the higher figure, so that `frac` stays a fraction.
GATHER_CEILING_G = 59.4
GATHER_CEILING_1GIB_G = 56.0  # the same microbenchmark on a 1 GiB table (nothing of it inside the Infinity Cache)
GATHER_CEILING_SRC = "profiles/r04/microbench_gather.jsonl (indep4/8, 64 MiB table: the fabric's request rate; 56.0 at 1 GiB, whole lines by quads 54.7)"
PMC_FALLBACK = "r05/pmc_query.json"


def cpu_model():
    """the host's CPU as /proc/cpuinfo names it (state nproc and CPU model beside the CPU baseline)"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--genomes", type=int, default=100)
    ap.add_argument("--genome-len", type=int, default=2_000_000)
    ap.add_argument("--snp-rate", type=float, default=0.01)
    ap.add_argument("--queries", type=int, default=125_000_000, help="queries per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries timed on the CPU baseline (0 = auto)")
    ap.add_argument("--no-secondary", action="store_true", help="N=1: only the headline, the roofline and the CPU baseline")
    ap.add_argument("--measure-ref-scan", action="store_true", help="also run the oracle's counting mode (bytes the reference's scan dereferences per query)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 counter passes (roofline from profiles/ if not stale)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the bitmap all_gather even with one rank (path check)")
    ap.add_argument("--replicate", choices=["rebuild", "broadcast"], default="broadcast",
                    help="N>1, how every rank gets the trie: rank 0 builds it and one RCCL broadcast replicates the image (default), "
                         "or each rank builds it from the same seeded input (no collective outside the bitmap gather)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes (this parent never touches a GPU) and
    return the worst exit code; rank 0's stdout (the JSON line) is this process's stdout."""
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------
def pan_index(k, genomes, genome_len, snp_rate, device, local_rank, build_here=True):
    """The pan-genome index through the device insert path; returns (bft or None, sorted distinct keys of the stored k-mers, ...)"""
    import torch
    from bloomfiltertrie_amd import BFT, workloads as W
    dev = torch.device("cuda", local_rank)
    pan = W.PanGenome(genomes, genome_len, snp_rate, 4242, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if build_here:
        bft = BFT(k, device=local_rank)
        for gid in range(genomes):
            bft.add_genome(f"genome_{gid}")
        keys, n_in = W.build_index(bft, pan, k)
    else:  # another rank builds the index; this one only needs the key table for its ground truth
        bft, n_in = None, 0
        keys = [W.unique_keys(W.keys_of(W.pack_windows(pan.genome(g), k))) for g in range(genomes)]
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    allk = W.union_of(keys)
    return bft, allk, n_in, t_build


def timed_launches(bft, dq_ptr, nq, bits_ptr, stream, reps):
    """reps launches of the presence kernel on a resident batch; mean kernel time (HIP events on the launch stream)"""
    import torch
    bft.query_presence_dev(dq_ptr, nq, bits_ptr, stream)
    torch.cuda.synchronize()
    bft.kernel_time(reset=True)
    for _ in range(reps):
        bft.query_presence_dev(dq_ptr, nq, bits_ptr, stream)
    torch.cuda.synchronize()
    ms, n = bft.kernel_time(reset=True)
    return ms / max(1, n)


def load_profile_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except Exception:
            return None
    return None


def roofline_block(pmc, nq, avg_ms, launches, kernel, kmer_bytes, live):
    """frac = HBM bytes per launch (hardware counters + the calibrated stream correction) / mean kernel time / 8 TB/s.  design = the bytes the
    layout needs per query (the packed k-mer, its answer bit, one 64-byte line of the k-mer hash); wasted = counter / design."""
    design = kmer_bytes + 0.125 + 64.0
    out = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": kernel,
           "avg_launch_ms": round(avg_ms, 4), "launches": int(launches), "design_bytes_per_query": round(design, 3),
           "pmc": "live (tools/pmc_live.py, this run)" if live else "profiles/" + PMC_FALLBACK}
    if not pmc or "hbm_bytes_per_query" not in pmc:
        out["pmc_stale"] = True
        out["pmc"] = (pmc or {}).get("error", "no counter figures for this library")
        return out
    per_q = pmc["hbm_bytes_per_query"]
    traffic = per_q * nq
    achieved = traffic / (avg_ms * 1e-3) / 1e9
    out.update({"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": round(traffic),
                "traffic_counted": round(pmc.get("fetch_bytes_tallied", 0) + pmc.get("write_bytes", 0)), "stream_correction": round(pmc.get("stream_correction_bytes", 0)),
                "hbm_bytes_per_query": per_q, "wasted": round(per_q / design, 3), "frac_is": "estimate (counters + calibrated stream correction), not a bound",
                "traffic_note": "traffic = traffic_counted (FETCH_SIZE x 1024: 64 B tallied per L2->fabric read request, exact for whole-line gathers, profiles/r03/pmc_microbench.json; "
                                "+ WRITE_SIZE x 1024) + stream_correction (half of the coalesced query stream, which the fetch counter tallies at 1/2: MI355X_MICROARCH.md HBM)",
                "kernel_us_under_pmc": pmc.get("kernel_us_under_pmc_mean"), "lib_source_hash": pmc.get("lib_source_hash")})
    mpq = pmc.get("l2_misses_per_query")
    if mpq:
        rate = mpq * nq / (avg_ms * 1e-3) / 1e9
        out["gather"] = {"l2_misses_per_query": mpq, "l2_requests_per_query": pmc.get("l2_requests_per_query"), "G_misses_per_s": round(rate, 1),
                         "ceiling_G_per_s": GATHER_CEILING_G, "frac": round(rate / GATHER_CEILING_G, 3), "ceiling_source": GATHER_CEILING_SRC,
                         "ceiling_1GiB_table_G_per_s": GATHER_CEILING_1GIB_G, "frac_of_1GiB_ceiling": round(rate / GATHER_CEILING_1GIB_G, 3),
                         "note": "two ceilings, both measured: the table of the headline index (0.67 GB) lies partly inside the 256 MiB Infinity Cache, so the true bound "
                                 "is between them; a value above 1 against the 1 GiB figure says exactly that"}
    return out


def pmc_for(workload_key, nq, kmer_bytes, allow_live, opts=(), stored_key=None):
    """counter figures for the query kernel on `workload_key` (tools/pmc_query.py name) with the options `opts`: live passes, else the
    committed file if it was collected on this library's sources"""
    from tools import pmc_live
    if allow_live:
        res = pmc_live.collect(workload_key, nq, 3, "k_query", opts=opts, kmer_bytes=kmer_bytes)
        if "error" not in res and "hbm_bytes_per_query" in res:
            return res, True
        err = res.get("error")
    else:
        err = "live collection disabled"
    stored = (load_profile_json(PMC_FALLBACK) or {}).get(stored_key or workload_key)
    if stored and stored.get("lib_source_hash") == pmc_live.source_hash() and stored.get("queries_per_launch") == nq:
        return stored, False
    return {"error": f"{err}; profiles/{PMC_FALLBACK} is for other sources"}, False


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
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

    from bloomfiltertrie_amd import BFT, synth as S, workloads as W
    from bloomfiltertrie_amd.dist import GatherPipeline, replicate_image

    genomes, nq, k = args.genomes, args.queries, args.k
    B = S.kmer_bytes(k)

    with BFT(k, device=local_rank) as warm:  # loads the code objects and the library sort kernels once (not part of any figure)
        wk = S.distinct(S.kmers_of(S.random_genome(120000, 5), k))
        warm.set_option("build_msd", 2)
        warm.insert_kmers(wk, 0)
        warm.build()
        warm.query_presence(wk[:1000])

    # ---- the index (replicated on every rank) through the product insertion path, and the resident query batch ----
    broadcast = use_dist and args.replicate == "broadcast"
    bft, allk, n_pairs_in, t_insert = pan_index(k, genomes, args.genome_len, args.snp_rate, device, local_rank, build_here=(rank == 0 or not broadcast))
    if broadcast:
        bft = replicate_image(bft, local_rank, src=0, always_copy=args.force_dist and world == 1)
    info = bft.info()
    build_times = bft.build_time() if (rank == 0 or not broadcast) else {}
    assert info["kmers"] == int(allk.numel()), (info["kmers"], int(allk.numel()))

    g = torch.Generator(device=device)
    g.manual_seed(99 + rank)
    dq, qk = W.presence_batch(allk, k, nq, g)
    stream = torch.cuda.current_stream().cuda_stream
    pipe = GatherPipeline(lambda buf: bft.query_presence_dev(dq.data_ptr(), nq, buf.data_ptr(), stream), ((nq + 63) // 64) * 8, world, device, use_dist)

    for _ in range(max(1, args.warmup) if args.warmup else 0):
        pipe.step()
    pipe.drain()
    torch.cuda.synchronize()
    bft.kernel_time(reset=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step()
    pipe.drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = bft.kernel_time(reset=True)
    dbits, gathered = pipe.last()  # the last step's answers
    if use_dist:
        assert torch.equal(gathered[rank * dbits.numel():(rank + 1) * dbits.numel()], dbits)
        if pipe.nbuf > 1 and pipe.steps > 1:
            assert torch.equal(pipe.bits[0], pipe.bits[1])  # every step answers the same batch
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- correctness of the measured batch: every answer of this rank's shard against ground truth ----
    truth_t = W.member(allk, qk)
    got_t = W.bits_to_bool(dbits, nq)
    parity_ok = bool((got_t == truth_t).all())
    n_present = int(got_t.sum())
    del truth_t, got_t
    if use_dist:
        okt = torch.tensor([1 if parity_ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        parity_ok = bool(okt.item())

    # ---- the other sharded queries under the same process group (only in the dist-path check's child: `--force-dist`): colour rows and branching,
    # answered shard by shard on the replicated handle, gathered, and compared with the handle's own unsharded answers ----
    dist_extra = None
    if args.force_dist:
        from bloomfiltertrie_amd.dist import query_branching_sharded, query_color_rows_sharded
        sample = np.ascontiguousarray(dq[: min(nq, 200_000)].cpu().numpy())
        cb, crows = query_color_rows_sharded(bft, sample, genomes)
        hb, hrows = bft.query_color_rows(sample)
        nb8 = (len(sample) + 7) // 8
        ok_rows = bool((cb[:nb8] == np.asarray(hb)[:nb8]).all() and (crows == np.asarray(hrows).reshape(crows.shape)).all())
        bb, bcounts = query_branching_sharded(bft, sample)
        hbb, hcounts = bft.query_branching(sample, with_counts=True)
        ok_br = bool((bb[:nb8] == np.asarray(hbb)[:nb8]).all() and (bcounts == np.asarray(hcounts)).all())
        dist_extra = {"color_rows": ok_rows, "branching": ok_br, "queries": int(len(sample))}

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    value = nq * world * args.steps / elapsed / 1e6
    fp = bft.footprint()
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
            "workload": f"BASELINE configs[3] per GPU: k={k} (stand-in for k=31: the reference requires k%9==0; k=31 beside it in `k31`), {genomes}-genome BFT "
                        f"resident in every GPU's HBM, {nq:.4g} batched presence queries per GPU (50% stored / 50% SNP mutants), query batch sharded, "
                        f"bitmaps all_gathered over xGMI for N>1",
            "name": "config4", "k": k, "genomes": genomes, "genome_len": args.genome_len, "snp_rate": args.snp_rate,
            "queries_per_gpu": nq, "distinct_kmers": info["kmers"], "pairs": info["pairs"],
            "trie": {x: info[x] for x in ("nodes", "ccs", "child_nodes", "prefixes", "uc_rows", "root_ccs", "colorsets", "image_bytes")},
            "parallelism": f"query-shard x{world}, trie replicated ({args.replicate if use_dist else 'single copy'})",
        },
        "parity_ok": parity_ok,
        "dist_extra_paths": dist_extra,
        "answers_checked_per_gpu": nq,
        "present_fraction": round(n_present / nq, 4),
        "build": {"insert_build_s_incl_torch": round(t_insert, 3), "M_pairs_per_s_incl_torch_packing": round(n_pairs_in / max(t_insert, 1e-9) / 1e6, 1),
                  "note": "NOT the library's build rate: 100 stream-ordered insert calls + bft_gpu_build TOGETHER WITH torch's window packing and per-genome key tables of this "
                          "script (tools/bench_insert.py times the library alone: profiles/r04/insert_config3.json)",
                  **{k_: round(v, 2) for k_, v in build_times.items() if not k_.startswith("_")}},
    }
    avg_ms = kern_ms / max(1, launches)
    secondary = world == 1 and not args.no_secondary

    # ---- secondary: the host-buffer entry point (H2D + kernel + D2H through bft_gpu_query_presence); never `value` ----
    if secondary:
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

    # ---- footprint: HBM bytes per part of the image against the .bft file of the same index (the reference's own measure of size) ----
    bft_path = None
    if world == 1:
        try:
            shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
            bft_path = os.path.join(shm, f"bench_{os.getpid()}.bft")
            t0 = time.perf_counter()
            bft.write_bft(bft_path)
            t_w = time.perf_counter() - t0
            fb = os.path.getsize(bft_path)
            img = info["image_bytes"]
            out["footprint"] = {"image_bytes": img, "bft_file_bytes": fb, "image_over_file": round(img / fb, 2),
                                "image_bytes_per_kmer": round(img / info["kmers"], 2), "file_bytes_per_kmer": round(fb / info["kmers"], 2),
                                "parts": fp, "write_bft_s": round(t_w, 2),
                                "note": "image = every array a query may touch (the index is its own store: no (k-mer, genome) pair list is kept)"}
            # the image as queries leave it ("compact_table", the default: the sorted table and the colour set per k-mer are not resident --
            # the k-mer hash holds both; write_bft just brought them back) against the image with the table kept; the same launches, timed
            try:
                bft.set_option("compact_table", 1)
                cimg, cfp = bft.info()["image_bytes"], bft.footprint()
                bits_c = torch.zeros_like(dbits)
                c_ms = timed_launches(bft, dq.data_ptr(), nq, bits_c.data_ptr(), stream, 5)
                same = bool(torch.equal(bits_c, dbits))
                del bits_c
                bft.set_option("compact_table", 0)
                timg = bft.info()["image_bytes"]
                out["footprint"].update({"image_bytes": cimg, "image_over_file": round(cimg / fb, 2), "image_bytes_per_kmer": round(cimg / info["kmers"], 2), "parts": cfp,
                                         "M_kmers_per_s": round(nq / c_ms / 1e3, 1), "same_bits": same,
                                         "kmer_hash_bytes_per_kmer": round(cfp["kmer_hash"] / info["kmers"], 2),
                                         "with_sorted_table": {"image_bytes": timg, "image_over_file": round(timg / fb, 2), "image_bytes_per_kmer": round(timg / info["kmers"], 2),
                                                               "note": "bft_gpu_set_option(compact_table, 0), or after rows / extraction / a merge / write_bft / the container walk "
                                                                       "brought the table back (it then stays until the next build)"}})
            finally:
                bft.set_option("compact_table", 1)
        except Exception as e:
            out["footprint"] = {"error": repr(e), "parts": fp}

    # ---- CPU baseline (oracle "port") on the SAME trie: the .bft file written above is loaded by the oracle's restatement of
    # read_BFT_Root and queried with its isKmerPresent loop on all host cores; a counting-mode pass gives the bytes the
    # reference's scan dereferences per query (SURVEY 8d's S) ----
    ref_scan = None
    if world == 1 and not args.no_cpu_baseline and bft_path and os.path.exists(bft_path):
        try:
            from oracle import oracle as O
            # the cores this process may run on: the affinity mask, cut by the cgroup's CPU quota when there is one
            cores_os = os.cpu_count() or 1
            cores_aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores_os
            quota = None
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    quota = float(q) / float(per)
            except Exception:
                pass
            cores = max(1, min(cores_aff, int(quota) if quota and quota >= 1 else cores_aff))
            t0 = time.time()
            orc = O.OracleBFT.load_bft(bft_path)
            orc.freeze()  # (the packed arrays of include/CC.h:34-67 the query loop runs on: built here, not inside the timed loop)
            t_load = time.time() - t0
            ns = args.cpu_sample or min(nq, 400_000 * cores)
            sample = np.ascontiguousarray(dq[:ns].cpu().numpy())
            orc.query_presence(sample[: min(ns, 200_000 * cores)], threads=cores)  # (first touch of the trie's pages by the threads that will read them)
            t0 = time.time()
            obits = orc.query_presence(sample, threads=cores)
            t_q = time.time() - t0
            scaling = {}
            for th in sorted({1, 8, 64, cores}):
                if th > cores:
                    continue
                n_th = min(ns, 1_500_000 * th)
                t0 = time.time()
                orc.query_presence(sample[:n_th], threads=th)
                scaling[str(th)] = round(n_th / (time.time() - t0) / 1e6, 3)
            orc.close()
            gpu_bits = dbits[: (ns + 7) // 8].cpu().numpy()
            tail = ns % 8
            same = bool((obits[: ns // 8] == gpu_bits[: ns // 8]).all()) and (tail == 0 or (obits[-1] ^ gpu_bits[-1]) & ((1 << tail) - 1) == 0)
            out["oracle_parity_ok"] = same
            out["cpu_baseline"] = {
                "value": round(ns / t_q / 1e6, 3), "unit": "M k-mers/s", "cores": cores, "kind": "port",
                "sample": f"first {ns} queries of the same batch; the oracle's isKmerPresent loop over {cores} threads on the trie the GPU built, read from the "
                          f".bft file it wrote by the oracle's restatement of read_BFT_Root ({t_load:.1f} s)",
                "cpu_model": cpu_model(), "cores_detail": {"os_cpu_count": cores_os, "sched_affinity": cores_aff, "cgroup_cpu_quota": quota, "threads_used": cores},
                "M_kmers_per_s_by_threads": scaling, "single_thread": scaling.get("1"),
                "note": "a reported baseline, not the target; the thread-scaling row says how far the host's memory system carries the loop",
            }
            if args.measure_ref_scan:  # the counting build of the oracle on the same file (another load): SURVEY 8d's S, live
                cnt = O.OracleBFT.load_bft(bft_path, count=True)
                nc = min(ns, 1_000_000)
                cbits, c = cnt.query_presence_count(sample[:nc])
                cnt.close()
                S_mean = c["bytes"] / nc
                ref_scan = B + 0.125 + S_mean
                out["reference_scan_bytes_per_query"] = {"total": round(ref_scan, 2), "kmer_in": B, "bit_out": 0.125, "trie_S": round(S_mean, 2),
                                                         "ccs_scanned": round(c["ccs_scanned"] / nc, 2), "levels": round(c["levels"] / nc, 3), "source": "this run"}
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    if bft_path and os.path.exists(bft_path):
        os.unlink(bft_path)

    # ---- roofline of the timed kernel: hardware counters of THIS library on THIS workload ----
    std = (genomes, args.genome_len, args.snp_rate, k) == (100, 2_000_000, 0.01, 27)
    if std:
        pmc, live = pmc_for("cfg4", nq, B, allow_live=(world == 1 and not args.no_pmc))
    else:
        pmc, live = {"error": "non-standard workload: no counter passes"}, False
    out["roofline"] = roofline_block(pmc, nq, avg_ms, launches, "k_query_kh" if build_times.get("kmer_hash_lines") else "k_query", B, live)
    if ref_scan is None and std:  # recorded by an earlier run of this bench with --measure-ref-scan (the oracle's counting mode on the same index)
        rec = load_profile_json("r03/ref_scan_bytes_config4.json")
        if rec:
            ref_scan = rec.get("total")
            out["reference_scan_bytes_per_query"] = dict(rec, source="profiles/r03/ref_scan_bytes_config4.json")
    if ref_scan and out["roofline"].get("hbm_bytes_per_query"):
        out["roofline"]["ref_scan_bytes_avoided"] = round(ref_scan / out["roofline"]["hbm_bytes_per_query"], 2)

    if secondary:
        # ---- the same index through the container walk, three ways, each with its own counters (bft_gpu_set_option names in `options`):
        #   pure         the kernels the north star names and nothing else: Bloom probe of the root's bit-sliced filters staged in LDS -> filter2 bit +
        #                rank-in-word -> cluster -> prefix entry -> suffix-group search / node UC (src/presenceNode.c:1284-1576, 1578-1821), every level
        #                through the containers: no k-mer hash, no derived root tables, no node prefix hash
        #   sorted_table the same walk with the derived root tables (rstart / rdir / rq) in front of the root's containers
        #   walk_hash    plain root groups looked up in their hashed form (one line of the k-mer hash, k_query6h), special prefixes walk
        # `roofline.frac` of the pure row prices the launch with SURVEY 8(d)'s algorithmic bytes -- B(k) + 1/8 + S, S = the bytes the reference's scan
        # dereferences per query on this index (the oracle's counting mode) -- so that it says how the kernel stands against the reference's own
        # traffic model; `counters` beside it is the HBM traffic the device layout really causes ----
        try:
            rows = [("pure", (("kmer_hash", 0), ("root_direct", 0), ("node_hash", 0)), "k_query", "cfg4_walk_pure"),
                    ("sorted_table", (("kmer_hash", 0),), "k_query", "cfg4_walk"),
                    ("walk_hash", (("walk_hash", 1),), "k_query6h", "cfg4_walk_hash")]
            defaults = {"kmer_hash": 1, "root_direct": 3, "node_hash": 1, "walk_hash": 0}
            cw = {}
            bits_w = torch.zeros_like(dbits)
            for label, opts, kern, stored in rows:
                try:
                    for nm, v in opts:
                        bft.set_option(nm, v)
                    bits_w.zero_()
                    ms_w = timed_launches(bft, dq.data_ptr(), nq, bits_w.data_ptr(), stream, 3)
                    row = {"value": round(nq / ms_w / 1e3, 3), "unit": "M k-mers/s", "ms_per_launch": round(ms_w, 4), "same_bits": bool(torch.equal(bits_w, dbits)),
                           "options": {nm: v for nm, v in opts}}
                    if std:
                        pmc_w, live_w = pmc_for("cfg4", nq, B, allow_live=not args.no_pmc, opts=tuple(f"{nm}={v}" for nm, v in opts), stored_key=stored)
                        rl = roofline_block(pmc_w, nq, ms_w, 3, kern, B, live_w)
                        if label != "walk_hash":
                            rl.pop("design_bytes_per_query", None)
                            rl.pop("wasted", None)
                        row["lines_per_query"] = (rl.get("gather") or {}).get("l2_misses_per_query")
                        if label == "pure" and ref_scan:
                            # achieved = algorithmic bytes per launch / launch time, algorithmic bytes = B(k) + 1/8 + S
                            ach = ref_scan * nq / (ms_w * 1e-3) / 1e9
                            row["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                               "traffic": rl.get("traffic"), "kernel": kern, "avg_launch_ms": round(ms_w, 4), "launches": 3,
                                               "algorithmic_bytes_per_query": ref_scan,
                                               "algorithmic_bytes_are": "SURVEY 8(d): B(k) + 1/8 + S, S = bytes the reference's presenceKmer/findCluster scan dereferences per query "
                                                                        "on this index (the oracle's counting mode, `reference_scan_bytes_per_query`)",
                                               "note": "a frac above what the counters say means the device layout (bit-sliced Bloom blocks: 2 loads instead of 2 per CC scanned; "
                                                       "rank in the filter2 word; fused prefix entries) moves fewer bytes than the reference's scan for the same answer",
                                               "counters": rl}
                        else:
                            row["roofline"] = rl
                    cw[label] = row
                finally:
                    for nm, _ in opts:
                        bft.set_option(nm, defaults[nm])
            cw["note"] = ("three forms of the container walk on the headline index and batch; `pure` = the kernels BASELINE.json's north star names (CC prefix match with popcnt "
                          "rank/select, UC suffix scan, Bloom probe of LDS-staged blocks); `sorted_table` adds the derived root tables; `walk_hash` answers plain root groups from "
                          "their hashed form.  The headline (`value`) is none of these: it is the k-mer hash alone (k_query_kh)")
            out["container_walk"] = cw
            del bits_w
        except Exception as e:
            out["container_walk"] = dict(out.get("container_walk") or {}, error=repr(e))
        bft.close()
        del dq, qk, allk
        torch.cuda.empty_cache()

        # ---- the index the north star names at the k it names: k = 31 (an extension: the reference rejects it, ground truth only) ----
        try:
            b31, allk31, n_in31, t_b31 = pan_index(31, genomes, args.genome_len, args.snp_rate, device, local_rank)
            g = torch.Generator(device=device)
            g.manual_seed(99)
            dq31, qk31 = W.presence_batch(allk31, 31, nq, g)
            bits31 = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=device)
            ms31 = timed_launches(b31, dq31.data_ptr(), nq, bits31.data_ptr(), stream, max(3, args.steps // 2))
            ok31 = bool((W.bits_to_bool(bits31, nq) == W.member(allk31, qk31)).all())
            i31 = b31.info()
            out["k31"] = {"value": round(nq / ms31 / 1e3, 3), "unit": "M k-mers/s", "ms_per_launch": round(ms31, 4), "queries": nq, "all_answers_checked": ok31,
                          "insert_build_s": round(t_b31, 3), "trie": {x: i31[x] for x in ("kmers", "pairs", "colorsets", "nodes", "ccs", "child_nodes", "image_bytes")},
                          "note": "the same 100-genome workload at k=31: rejected by the reference (k % 9 != 0), no oracle exists; every answer checked against set membership"}
            b31.close()
            del dq31, qk31, bits31, allk31
            torch.cuda.empty_cache()
        except Exception as e:
            out["k31"] = {"error": repr(e)}

        # ---- BASELINE configs[1]: 10-genome index (the image fits the 256 MiB Infinity Cache), 10^8 queries ----
        b2 = None
        try:
            b2, allk2, n_in2, t_b2 = pan_index(k, 10, args.genome_len, args.snp_rate, device, local_rank)
            g = torch.Generator(device=device)
            g.manual_seed(99)
            n2 = 100_000_000
            dq2, qk2 = W.presence_batch(allk2, k, n2, g)
            bits2 = torch.zeros(((n2 + 63) // 64) * 8, dtype=torch.uint8, device=device)
            ms2 = timed_launches(b2, dq2.data_ptr(), n2, bits2.data_ptr(), stream, max(3, args.steps // 2))
            ok2 = bool((W.bits_to_bool(bits2, n2) == W.member(allk2, qk2)).all())
            i2 = b2.info()
            out["config2"] = {"value": round(n2 / ms2 / 1e3, 3), "unit": "M k-mers/s", "ms_per_launch": round(ms2, 4), "queries": n2, "all_answers_checked": ok2,
                              "trie": {x: i2[x] for x in ("kmers", "pairs", "colorsets", "nodes", "ccs", "child_nodes", "image_bytes")},
                              "note": "BASELINE configs[1]: 10-genome BFT resident in HBM, 10^8 batched presence queries"}
            del dq2, qk2, bits2
        except Exception as e:
            out["config2"] = {"error": repr(e)}

        # ---- sequence queries on the 10-genome index, device-resident reads ----
        try:
            n_reads, rl, g10 = 200_000, 150, 10
            pan = W.PanGenome(g10, args.genome_len, args.snp_rate, 4242, device)
            gs = torch.stack([pan.genome(gi) for gi in range(g10)])
            gen = torch.Generator(device=device)
            gen.manual_seed(11)
            src = torch.randint(0, g10, (n_reads,), generator=gen, device=device)
            start = torch.randint(0, args.genome_len - rl, (n_reads,), generator=gen, device=device)
            reads = gs[src[:, None], start[:, None] + torch.arange(rl, device=device)[None, :]]
            d_blob = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)[reads.long()].reshape(-1).contiguous()
            d_off = (torch.arange(n_reads + 1, dtype=torch.int64, device=device) * rl).contiguous()
            rowbytes = (g10 + 7) // 8
            d_rows = torch.zeros((n_reads, rowbytes), dtype=torch.uint8, device=device)
            call = lambda: b2.query_sequences_dev(d_blob.data_ptr(), d_off.data_ptr(), n_reads, n_reads * rl, 1.0, d_rows.data_ptr(), False, stream)
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms_s = e0.elapsed_time(e1) / 5
            rows = np.unpackbits(d_rows.cpu().numpy(), axis=1, bitorder="little")[:, :g10]
            # every k-mer of a read cut out of genome g is in genome g: bit g must be set at threshold 1.0; the host-buffer call must agree
            own = bool(rows[np.arange(n_reads), src.cpu().numpy()].all())
            blob_h = d_blob.cpu().numpy().reshape(n_reads, rl)
            host = b2.query_sequences([bytes(r) for r in blob_h[:2000]], 1.0)
            same = host == [np.flatnonzero(r).tolist() for r in rows[:2000]]
            out["sequence_queries"] = {"value": round(n_reads / ms_s / 1e3, 2), "unit": "M reads/s", "M_kmers_per_s": round(n_reads * (rl - k + 1) / ms_s / 1e3, 1),
                                       "reads": n_reads, "read_len": rl, "threshold": 1.0, "ms": round(ms_s, 3), "source_genome_bit_set": own,
                                       "host_call_agrees_on_2000": bool(same),
                                       "note": "bft_gpu_query_sequences_dev on the 10-genome index: reads resident in HBM, error-free substrings of the indexed genomes"}
        except Exception as e:
            out["sequence_queries"] = {"error": repr(e)}
        if b2 is not None:
            b2.close()
        torch.cuda.empty_cache()

        # ---- BASELINE configs[2]: the insertKmers build path (src/insertNode.c:18-36), 100 genomes / 2x10^8 pairs from HBM-resident batches: the protocol
        # of tools/bench_insert.py (every batch generated before the clock starts; timed region = 100 stream-ordered insert calls + bft_gpu_build), with
        # the GPU time and the algorithmic bytes of every stage (bft_gpu_build_stages) and the oracle's insertKmers timed beside it ----
        try:
            from tools import bench_insert
            ia = argparse.Namespace(k=k, genomes=genomes, genome_len=args.genome_len, snp_rate=args.snp_rate, sample=400_000, reserve=True, sync_inserts=False,
                                    add_genome=False, opt=[], stages=True, warm_pool=True, cpu_baseline=0 if args.no_cpu_baseline else 8)  # (warm_pool: the same build once before,
            # on a handle that is closed again -- the steady state of a process that builds repeatedly: every block size of this build is in the library's cache)
            out["insert"] = bench_insert.measure(ia)
        except Exception as e:
            out["insert"] = {"error": repr(e)}

    # ---- the N > 1 path with one rank, in a child process (this one never initialised RCCL): process group over RCCL, the image blob packed and
    # broadcast (always_copy: the rank unpacks the received blob instead of keeping its own handle), presence bitmaps all_gathered with the overlap
    # pipeline, gathered shard == own bits, every answer checked -- so that the code an 8-GPU run takes is executed in every bench run ----
    if secondary and not args.force_dist:
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--force-dist", "--steps", "3", "--warmup", "1", "--genomes", "10", "--queries", "20000000",
                   "--no-secondary", "--no-cpu-baseline", "--no-pmc"]
            env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            t0 = time.perf_counter()
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            line = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
            if r.returncode == 0 and line:
                d = json.loads(line[-1])
                ex = d.get("dist_extra_paths") or {}
                out["dist_path_check"] = {"ok": bool(d.get("parity_ok")) and bool(ex.get("color_rows")) and bool(ex.get("branching")),
                                          "paths": {"presence (overlapped all_gather)": bool(d.get("parity_ok")), "color_rows (sharded, rows gathered)": bool(ex.get("color_rows")),
                                                    "branching (sharded, counts gathered)": bool(ex.get("branching"))},
                                          "backend": "nccl (RCCL)", "world_size": 1, "replicate": "broadcast of the packed image, unpacked (always_copy)",
                                          "gather": "all_gather of the presence bitmaps, overlapped with the next step's kernel (GatherPipeline)",
                                          "value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"), "answers_checked": d.get("answers_checked_per_gpu"),
                                          "workload": "10-genome index, 2x10^7 queries, 3 steps", "seconds": round(time.perf_counter() - t0, 1),
                                          "note": "one rank: what RCCL does between ranks over xGMI is NOT exercised; the process-group, blob, unpack and gather code is"}
            else:
                out["dist_path_check"] = {"ok": False, "rc": r.returncode, "stderr_tail": r.stderr.decode(errors="replace")[-400:]}
        except Exception as e:
            out["dist_path_check"] = {"ok": False, "error": repr(e)}

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
