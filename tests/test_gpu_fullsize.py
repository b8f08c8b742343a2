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
    from bloomfiltertrie_amd.workloads import make_queries_on_device
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


def test_host_entry_points_across_their_chunk_boundaries():
    """The host-buffer entry points work in chunks (2^26 k-mers for presence / branching, 2^24 for colour lists and row
    locations, 2^22 for colour rows): batches one chunk plus a ragged tail long, checked against ground truth and against
    each other, so that bitmap bytes, offsets and rows line up across the boundary."""
    from bloomfiltertrie_amd import BFT
    k = 27
    anc = S.random_genome(200000, 3)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.02, 40 + g), k)) for g in range(3)]
    t = BFT(k)
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
    allk = S.distinct(np.concatenate(gk))
    base = np.concatenate([allk, S.snp_mutants(allk, k, 2)])
    rng = np.random.default_rng(1)

    def batch(n):
        return np.ascontiguousarray(base[rng.integers(0, len(base), n)])

    # presence: 2^26 + 777
    n = (1 << 26) + 777
    q = batch(n)
    bits = t.query_presence(q)
    truth = S.member(q, allk)
    assert (S.from_bits(bits, n).astype(bool) == truth).all()
    # colour lists and row locations: 2^24 + 13 (a prefix of the same batch)
    m = (1 << 24) + 13
    b2, off, ids = t.query_colors(q[:m])
    assert (S.from_bits(b2, m).astype(bool) == truth[:m]).all()
    assert off[0] == 0 and (np.diff(off.astype(np.int64)) >= 0).all() and off[-1] == len(ids)
    assert ((np.diff(off.astype(np.int64)) > 0) == truth[:m]).all()
    b3, rows, sets = t.query_rows(q[:m])
    assert (b3 == b2).all() and ((rows != 0xFFFFFFFF) == truth[:m]).all()
    stored, cs = t.extract()
    pr = truth[:m]
    assert (stored[rows[pr]] == q[:m][pr]).all() and (cs[rows[pr]] == sets[pr]).all()
    # sizes of the id lists around the 2^24 boundary == sizes of the colour sets found by query_rows
    set_size = {c: len(t.colorset(c)) for c in np.unique(sets[pr]).tolist()}
    for i in list(range((1 << 24) - 50, m)):
        assert int(off[i + 1] - off[i]) == (set_size[int(sets[i])] if pr[i] else 0)
    # colour rows: 2^22 + 5
    r = (1 << 22) + 5
    b4, crow = t.query_color_rows(q[:r])
    assert (S.from_bits(b4, r).astype(bool) == truth[:r]).all()
    unp = np.unpackbits(crow, axis=1, bitorder="little")[:, :3]
    assert (unp.any(axis=1) == truth[:r]).all()
    for i in list(range((1 << 22) - 20, r)) + list(range(0, 20)):
        have = ids[int(off[i]):int(off[i + 1])].tolist()
        assert np.flatnonzero(unp[i]).tolist() == have
    # branching: same boundary as presence; idempotent and equal to the device-side popcount on a slice
    bb = t.query_branching(q)
    bb2 = t.query_branching(q[: (1 << 26)])
    assert (bb[: (1 << 23)] == bb2).all()


def test_more_than_2_31_pairs_on_one_gpu():
    """No bound on the (k-mer, genome) pairs of an index (the reference has none, src/insertNode.c:18-36; rounds 1-2 stopped at 2^31 - 1):
    1100 genomes of 2 Mbp = 2.2x10^9 pairs, 5x10^8 distinct k-mers, through the ordinary insert calls -- the log is merged into the
    index every 2^30 pairs (bft_merge.hip).  Checked: the pair and k-mer counts against torch's own sort of every key, presence and the
    source genome's colour bit on samples of three genomes, absence / presence of mutants against the sorted keys."""
    import torch
    from bloomfiltertrie_amd import BFT, workloads as W
    k, G, glen = 27, 1100, 2_000_000
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    pan = W.PanGenome(G, glen, 0.01, 77, dev)
    t = BFT(k)
    keys, n_pairs = [], 0
    for g in range(G):
        packed = W.pack_windows(pan.genome(g), k)
        t.insert_kmers_dev_async(packed.data_ptr(), packed.shape[0], g, stream)
        kg = W.unique_keys(W.keys_of(packed))
        n_pairs += int(kg.numel())
        keys.append(kg)
        del packed
    t.build()
    info = t.info()
    assert n_pairs > (1 << 31) and info["pairs"] == n_pairs and info["pending_pairs"] == 0 and info["genomes"] == G
    # (torch sorts at most 2^31 - 1 elements at a time: the union of the per-genome key tables, four parts first)
    parts = [torch.unique(torch.cat(keys[a::4])) for a in range(4)]
    del keys
    allk = torch.unique(torch.cat(parts))
    del parts
    assert info["kmers"] == int(allk.numel())
    rowbytes = (G + 7) // 8
    for g in (0, 537, G - 1):
        packed = W.pack_windows(pan.genome(g), k)[::53].contiguous()
        n = packed.shape[0]
        bits = torch.zeros(((n + 63) // 64) * 8, dtype=torch.uint8, device=dev)
        rows = torch.zeros((n, rowbytes), dtype=torch.uint8, device=dev)
        scratch = torch.zeros(n, dtype=torch.int32, device=dev)
        t.query_color_rows_dev(packed.data_ptr(), n, bits.data_ptr(), rows.data_ptr(), scratch.data_ptr(), stream)
        torch.cuda.synchronize()
        assert bool(W.bits_to_bool(bits, n).all())
        assert bool(((rows[:, g // 8] >> (g % 8)) & 1).bool().all())
        # colour-set sizes are plausible: an unmutated window is shared by most genomes, and no set is empty
        sizes = torch.zeros(n, dtype=torch.int32, device=dev)
        for b in range(rowbytes):
            col = rows[:, b].int()
            for j in range(8):
                sizes += (col >> j) & 1
        assert int(sizes.min()) >= 1 and int(sizes.max()) > G // 2
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    nq = 20_000_000
    idx = torch.randint(0, allk.numel(), (nq,), generator=gen, device=dev)
    qk = W.snp_mutate_keys(allk[idx], k, 0.5, gen)
    dq = W.packed_of(qk, k)
    bits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
    t.query_presence_dev(dq.data_ptr(), nq, bits.data_ptr(), stream)
    torch.cuda.synchronize()
    p = torch.searchsorted(allk, qk).clamp(max=allk.numel() - 1)
    assert bool((W.bits_to_bool(bits, nq) == (allk[p] == qk)).all())
    t.close()
