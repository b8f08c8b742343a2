"""BASELINE.json full sizes on the GPU box, checked through size-independent properties (the oracle cannot answer
10^8 queries in test time):
 - config 2: 10-genome k=27 trie, 10^8 resident queries: every present k-mer of the index answers 1 (popcount == n
   for an all-present batch), single-SNP mutants of present k-mers and the full batch against a torch searchsorted
   ground truth, idempotence across launches, device API == host API on a slice;
 - set algebra: presence of the union batch == OR of the per-genome presences for tries built per genome."""
import numpy as np
import pytest

from bloomfiltertrie_amd import synth as S

pytestmark = pytest.mark.gpu


def _keys_t(packed_t):
    import torch
    n, nb = packed_t.shape
    pad = torch.zeros((n, 8), dtype=torch.uint8, device=packed_t.device)
    pad[:, :nb] = packed_t
    return pad.view(torch.int64).reshape(n)


def test_config2_full_batch_properties():
    import torch
    from bloomfiltertrie_amd import BFT
    from bench import make_queries_on_device
    k, nq = 27, 100_000_000
    anc = S.random_genome(2_000_000, 1234)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 1000 + g), k)) for g in range(10)]
    t = BFT(k)
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
    t.build()
    union = S.distinct(np.concatenate(gk))
    assert t.info()["kmers"] == len(union)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    # (1) an all-present batch: 10^8 samples of the union
    U = torch.from_numpy(union).to(dev)
    idx = torch.randint(0, U.shape[0], (nq,), device=dev)
    dq = U[idx]
    dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    cnt = int(torch.from_numpy(np.unpackbits(dbits.cpu().numpy(), bitorder="little")[:nq]).sum())
    assert cnt == nq
    del dq, idx
    # (2) the bench batch (50 % present / 50 % SNP mutants) against searchsorted ground truth, all 10^8 queries
    dq = make_queries_on_device(union, k, nq, 99, dev)
    dbits.zero_()
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    first = dbits.clone()
    ukeys, _ = torch.sort(_keys_t(U))
    qkeys = _keys_t(dq)
    pos = torch.searchsorted(ukeys, qkeys).clamp(max=ukeys.numel() - 1)
    truth = (ukeys[pos] == qkeys).cpu().numpy()
    got = np.unpackbits(first.cpu().numpy(), bitorder="little")[:nq].astype(bool)
    assert (got == truth).all()
    # (3) idempotence
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(first, dbits)
    # (4) host API == device API on a slice that is not 64-aligned in length
    ns = 1_000_003
    hb = t.query_presence(dq[:ns].cpu().numpy())
    assert (np.unpackbits(hb, bitorder="little")[:ns] == got[:ns]).all()


def test_union_is_or_of_genomes():
    from bloomfiltertrie_amd import BFT
    k = 36
    anc = S.random_genome(300000, 8)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.02, 70 + g), k)) for g in range(4)]
    whole = BFT(k)
    parts = []
    for g, km in enumerate(gk):
        whole.insert_kmers(km, g)
        p = BFT(k)
        p.insert_kmers(km, 0)
        parts.append(p)
    allk = S.distinct(np.concatenate(gk))
    q = np.concatenate([allk[::3], S.snp_mutants(allk[::5], k, 1)])
    w = S.from_bits(whole.query_presence(q), len(q))
    acc = np.zeros(len(q), bool)
    bits, rows = whole.query_color_rows(q)
    unp = np.unpackbits(rows, axis=1, bitorder="little")[:, :4].astype(bool)
    for g, p in enumerate(parts):
        pg = S.from_bits(p.query_presence(q), len(q))
        assert (pg == unp[:, g]).all()  # colour g of the whole index == presence in genome g's own index
        acc |= pg
    assert (w == acc).all()
