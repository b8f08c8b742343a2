#!/usr/bin/env python3
"""Presence throughput against k on the config-2 generator (10 genomes of 2 Mbp, 1 % SNPs; 10^8 resident queries, 50 % present /
50 % SNP mutants; half / a quarter as many beyond k = 32 / 64), every answer of a 500 000-query slice checked against set membership.
k % 9 != 0 are the extension; k = 72 / 99 / 126 are the three- and four-word keys (one to four slots per line of the k-mer hash)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from bloomfiltertrie_amd import BFT, synth as S
    from bloomfiltertrie_amd.workloads import make_queries_on_device
    dev = torch.device("cuda", 0)
    nq = 100_000_000
    out = []
    ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (18, 21, 27, 31, 32, 36, 45, 54, 63, 64, 72, 99, 126)
    load = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # "kmer_hash_load" (0: the default)
    for k in ks:
        anc = S.random_genome(2_000_000, 1234)
        gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 1000 + g), k)) for g in range(10)]
        t = BFT(k)
        if load:
            t.set_option("kmer_hash_load", load)
        for g, km in enumerate(gk):
            t.insert_kmers(km, g)
        t.build()
        union = S.distinct(np.concatenate(gk))
        n = nq if k <= 32 else (nq // 2 if k <= 64 else nq // 4)
        dq = make_queries_on_device(union, k, n, 5, dev)
        dbits = torch.zeros(((n + 63) // 64) * 8, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        t.query_presence_dev(dq.data_ptr(), n, dbits.data_ptr(), stream)
        torch.cuda.synchronize()
        nv = 500_000
        ok = bool((S.from_bits(dbits[: (nv + 7) // 8].cpu().numpy(), nv) == S.member(dq[:nv].cpu().numpy(), union)).all())
        t.kernel_time(reset=True)
        for _ in range(5):
            t.query_presence_dev(dq.data_ptr(), n, dbits.data_ptr(), stream)
        torch.cuda.synchronize()
        ms, cnt = t.kernel_time(reset=True)
        info = t.info()
        bt = t.build_time()
        out.append({"k": k, "kmer_hash_load": load or 55, "queries": n, "ms": round(ms / cnt, 3), "G_kmers_per_s": round(n / (ms / cnt) / 1e6, 2), "ok": ok, "kmers": info["kmers"],
                    "nodes": info["nodes"], "image_MB": round(info["image_bytes"] / 1e6, 1), "kh_slots": int(bt["kmer_hash_slots"]), "kh_maxd": int(bt["kmer_hash_maxd"]), "kh_overflow": int(bt["kmer_hash_overflow"]),
                    "kh_bytes_per_kmer": round(t.footprint()["kmer_hash"] / max(1, info["kmers"]), 2)})
        print(json.dumps(out[-1]), flush=True)
        t.close()
        del dq, dbits
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
