#!/usr/bin/env python3
"""Direct kernel vs prefix-bucketed batches (bft_kernels_bucket.h) on the config-2 and config-4 indexes (and k=63 / deep tries):
time per launch for each bucket width, every answer checked against ground truth.
usage: bench_bucket.py [--workloads cfg2,cfg4,cfg4k31] [--bits 0,6,8,10] [--reps 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="cfg2,cfg4")
    ap.add_argument("--bits", default="0,6,8,10")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--queries", type=int, default=0)
    ap.add_argument("--root-direct", default="1", help="comma list of 0/1: root level through the derived direct table")
    ap.add_argument("--wgs", default="0", help="comma list of query_wgs_per_cu settings (0 = automatic)")
    ap.add_argument("--group-hash", default="1", help="comma list of 0/1: suffix groups through their hashed form")
    args = ap.parse_args()
    import torch
    from bloomfiltertrie_amd import BFT, workloads as W
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for wl in args.workloads.split(","):
        k = 31 if wl.endswith("k31") else 27
        genomes, nq = (10, 100_000_000) if wl.startswith("cfg2") else (100, 125_000_000)
        nq = args.queries or nq
        pan = W.PanGenome(genomes, 2_000_000, 0.01, 4242, dev)
        t = BFT(k)
        keys, _ = W.build_index(t, pan, k)
        allk = W.union_of(keys)
        del keys
        g = torch.Generator(device=dev)
        g.manual_seed(99)
        dq, qk = W.presence_batch(allk, k, nq, g)
        truth = W.member(allk, qk)
        dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
        info = t.info()
        for wg, gh, rd, bits in [(int(w_), int(g_), int(r), int(x)) for w_ in args.wgs.split(",") for g_ in args.group_hash.split(",") for r in args.root_direct.split(",")
                                 for x in args.bits.split(",")]:
            t.set_option("query_wgs_per_cu", wg)
            t.set_option("group_hash", gh)
            t.set_option("root_direct", rd)
            t.set_option("query_bucket_bits", bits)
            dbits.zero_()
            t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
            torch.cuda.synchronize()
            ok = bool((W.bits_to_bool(dbits, nq) == truth).all())
            t.kernel_time(reset=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            kms, kn = t.kernel_time(reset=True)
            print(json.dumps({"workload": wl, "k": k, "bucket_bits": bits, "wgs": wg, "root_direct": rd, "group_hash": gh, "hashed_groups": t.build_time()["hashed_groups"], "unhashed_groups": t.build_time()["unhashed_groups"], "ms": round(ms, 3), "ms_events_in_lib": round(kms / max(1, kn), 3), "G_kmers_per_s": round(nq / ms / 1e6, 2),
                              "all_answers_ok": ok, "image_bytes": info["image_bytes"], "kmers": info["kmers"], "queries": nq}), flush=True)
        t.close()
        del dq, qk, truth, dbits, allk
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
