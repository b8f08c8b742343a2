#!/usr/bin/env python3
"""Claimed chunks against the static split of the k-mer hash kernels on ragged batch sizes around 2^25 k-mers: presence bits, colour-set
ids (the out32 path behind colour rows) and branching answers must be identical.  usage: soak_claims.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, synth as S  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(5)
for r in range(rounds):
    k = int(rng.choice([27, 31, 45, 63]))
    anc = S.random_genome(60000, 10 + r)
    t = BFT(k)
    for g in range(9):
        t.insert_kmers(S.distinct(S.kmers_of(S.mutate(anc, 0.02, 100 * r + g), k)), g)
    t.build()
    ek, _ = t.extract()
    base = np.concatenate([ek, S.snp_mutants(ek, k, r)])
    n = (1 << 25) + int(rng.integers(0, 5000))
    idx = torch.randint(0, len(base), (n,), device=dev)
    dq = torch.from_numpy(base).to(dev)[idx].contiguous()
    nb = 1 << 22 if r % 2 else n  # branching: claims from 4 x 10^6 k-mers on
    res = []
    for mode, chunk in ((0, 4), (1, 4), (1, int(rng.choice([1, 2, 3, 5, 7, 16, 64])))):
        t.set_option("query_dynamic", mode)
        t.set_option("query_chunk", chunk)
        bits = torch.zeros(((n + 63) // 64) * 8, dtype=torch.uint8, device=dev)
        rows = torch.zeros((n, 2), dtype=torch.uint8, device=dev)
        scratch = torch.empty(n, dtype=torch.int32, device=dev)
        t.query_color_rows_dev(dq.data_ptr(), n, bits.data_ptr(), rows.data_ptr(), scratch.data_ptr(), st)
        bb = torch.zeros(((nb + 63) // 64) * 8, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(nb, dtype=torch.uint8, device=dev)
        t.query_branching_dev(dq.data_ptr(), nb, bb.data_ptr(), cnt.data_ptr(), st)
        torch.cuda.synchronize()
        res.append((bits, rows, bb, cnt, chunk))
    for other in res[1:]:
        assert all(torch.equal(a, b) for a, b in zip(res[0][:4], other[:4])), (k, n, other[4])
    truth = torch.from_numpy(np.concatenate([np.ones(len(ek), bool), S.member(base[len(ek):], ek)])).to(dev)[idx]
    got = torch.from_numpy(np.unpackbits(res[1][0].cpu().numpy(), bitorder="little")[:n].astype(bool)).to(dev)
    assert bool((got == truth).all())
    print(f"round {r}: k={k} n={n} branching n={nb} chunks 4/{res[2][4]} ok", flush=True)
    t.close()
print("claims OK")
