"""BASELINE.json configs 3, 4 (per-GPU share) and 5 at their FULL sizes on one MI355X, every answer checked against ground
truth computed with torch from the per-genome sorted key tables (the oracle cannot build these in test time; the BFT is
an exact index, so presence == set membership and colour set == the genomes that inserted the k-mer).

 config 3  insertKmers build path: 100 genomes x 2 Mbp (~2x10^8 (k-mer, genome) pairs), counts + a 2x10^6-query
           presence / colour-row sample
 config 4  k = 27 (reference-compatible stand-in) AND k = 31 (the k the metric names; extension): 100-genome index,
           10^9 / 8 = 1.25x10^8 presence queries, ALL answers checked, host bitmap == device bitmap on a slice
 config 5  k = 63, 2000 colours x 20 kbp: -query_branching (10^7 k-mers: bit == counts rule, counts vs neighbours'
           membership on a sample), presence of all 10^7, colour rows of 4x10^6 k-mers (1 GB) vs the inserting genomes
 f-4       sequence queries: 10^6 reads x 150 nt on the 10-genome index, device-resident, per-genome counts of a 2x10^4-read
           slice against set membership of every k-mer position
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    return torch, torch.device("cuda", 0)


def _gen(torch, dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return g


def _colour_truth(torch, per_genome_keys, qk, W):
    exp = torch.zeros((qk.shape[0], len(per_genome_keys)), dtype=torch.bool, device=qk.device)
    for gid, gkeys in enumerate(per_genome_keys):
        exp[:, gid] = W.member(gkeys, qk)
    return exp


def test_config3_insert_build_full_size(torch_dev):
    torch, dev = torch_dev
    from bloomfiltertrie_amd import BFT, workloads as W
    k = 27
    pan = W.PanGenome(100, 2_000_000, 0.01, 4242, dev)
    t = BFT(k)
    keys, n_in = W.build_index(t, pan, k)
    info = t.info()
    allk = W.union_of(keys)
    pairs = sum(int(x.numel()) for x in keys)
    assert n_in == 100 * (2_000_000 - k + 1) and n_in > 1.9e8
    assert info["kmers"] == int(allk.numel()) and info["pairs"] == pairs and info["genomes"] == 100
    assert info["pending_pairs"] == 0 and info["child_nodes"] >= 0
    # sample: half stored k-mers, half uniform random ones; presence + colour rows against the inserting genomes
    ns = 2_000_000
    g = _gen(torch, dev, 7)
    idx = torch.randint(0, allk.numel(), (ns // 2,), generator=g, device=dev)
    qk = torch.cat([allk[idx], torch.randint(0, 1 << (2 * k), (ns - ns // 2,), generator=g, device=dev, dtype=torch.int64)])
    q = W.packed_of(qk, k).cpu().numpy()
    bits, rows = t.query_color_rows(q)
    got = torch.from_numpy(np.unpackbits(rows, axis=1, bitorder="little")[:, :100].astype(bool)).to(dev)
    exp = _colour_truth(torch, keys, qk, W)
    assert bool((got == exp).all())
    pres = torch.from_numpy(np.unpackbits(bits, bitorder="little")[:ns].astype(bool)).to(dev)
    assert bool((pres == exp.any(dim=1)).all()) and int(pres[: ns // 2].sum()) == ns // 2
    # incremental: re-inserting genome 0 changes nothing, a new genome 100 adds exactly its pairs
    p0 = W.pack_windows(pan.genome(0), k)
    t.insert_kmers_dev(p0.data_ptr(), p0.shape[0], 0)
    extra = W.pack_windows(W.PanGenome(1, 2_000_000, 0.01, 999, dev).genome(0), k)
    t.insert_kmers_dev(extra.data_ptr(), extra.shape[0], 100)
    t.build()
    info2 = t.info()
    ek = W.unique_keys(W.keys_of(extra))
    assert info2["pairs"] == pairs + int(ek.numel()) and info2["kmers"] == int(W.unique_keys(torch.cat([allk, ek])).numel())
    t.close()


@pytest.mark.parametrize("k", [27, 31])
def test_config4_per_gpu_share_full_size(torch_dev, k):
    torch, dev = torch_dev
    from bloomfiltertrie_amd import BFT, workloads as W
    nq = 125_000_000
    pan = W.PanGenome(100, 2_000_000, 0.01, 4242, dev)
    t = BFT(k)
    keys, _ = W.build_index(t, pan, k)
    allk = W.union_of(keys)
    del keys
    info = t.info()
    assert info["kmers"] == int(allk.numel()) and info["genomes"] == 100 and info["image_bytes"] > 256 << 20  # beyond the Infinity Cache
    dq, qk = W.presence_batch(allk, k, nq, _gen(torch, dev, 99))
    dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    truth = W.member(allk, qk)
    got = W.bits_to_bool(dbits, nq)
    assert bool((got == truth).all())            # every one of the 1.25x10^8 answers
    assert 0.5 < float(truth.float().mean()) < 0.7
    first = dbits.clone()
    for opts in ({"query_wgs_per_cu": 1, "query_probe": 4}, {"query_wgs_per_cu": 2, "query_probe": 8}):  # launch options never change answers
        for name, v in opts.items():
            t.set_option(name, v)
        dbits.zero_()
        t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
        torch.cuda.synchronize()
        assert torch.equal(first, dbits)
    # The k-mer hash kernel claims its blocks of k-mers from a per-stream counter at this size ("query_dynamic", default): the static
    # split gives the same bits; two streams at once -- a counter pair each -- and ragged sizes around the chunk of 2048 k-mers too;
    # the counters are left zeroed, so the launches that follow are whole again.
    t.set_option("query_dynamic", 0)
    dbits.zero_()
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(first, dbits)
    t.set_option("query_dynamic", 1)
    s2 = torch.cuda.Stream(device=dev)
    n2 = (1 << 25) + 2049 + 77
    other = torch.zeros(((n2 + 63) // 64) * 8, dtype=torch.uint8, device=dev)
    for chunk in (4, 16, 3):
        t.set_option("query_chunk", chunk)
        for _ in range(3):
            dbits.zero_()
            other.zero_()
            torch.cuda.synchronize()
            t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
            t.query_presence_dev(dq.data_ptr(), n2, other.data_ptr(), s2.cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(first, dbits)
            assert bool((W.bits_to_bool(other, n2) == truth[:n2]).all())
    t.set_option("query_chunk", 4)
    ns = 3_000_001  # host entry point on a ragged slice
    hb = t.query_presence(dq[:ns].cpu().numpy())
    assert (np.unpackbits(hb, bitorder="little")[:ns].astype(bool) == got[:ns].cpu().numpy()).all()
    t.close()


def test_config5_branching_and_colour_rows_full_size(torch_dev):
    torch, dev = torch_dev
    from bloomfiltertrie_amd import BFT, _lib as L, synth as S, workloads as W
    k, G, glen, nq = 63, 2000, 20000, 10_000_000
    pan = W.PanGenome(G, glen, 0.01, 77, dev)
    t = BFT(k)
    keys, n_in = W.build_index(t, pan, k)
    allk = W.union_of(keys)
    info = t.info()
    assert n_in == G * (glen - k + 1)
    assert info["kmers"] == allk.shape[0] and info["pairs"] == sum(x.shape[0] for x in keys) and info["genomes"] == G
    # queries: stored k-mers, half of them with one SNP
    g = _gen(torch, dev, 1)
    idx = torch.randint(0, allk.shape[0], (nq,), generator=g, device=dev)
    sel = allk[idx]
    stored = torch.cat([sel[:, 1:2], sel[:, 0:1]], dim=1).contiguous().view(torch.uint8).reshape(nq, 16)  # back to byte order
    dq = W.snp_mutate_packed(stored, k, 0.5, g)
    del stored, sel, idx
    qk = W.keys_of(dq)
    stream = torch.cuda.current_stream().cuda_stream
    dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
    # ---- presence of all 10^7 ----
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    truth = W.member(allk, qk)
    assert bool((W.bits_to_bool(dbits, nq) == truth).all())
    # ---- branching: bits and counts for all 10^7; the bit is (successors > 1 or predecessors > 1) ----
    dcnt = torch.zeros(nq, dtype=torch.uint8, device=dev)
    bbits = torch.zeros_like(dbits)
    L.check(t._lib.bft_gpu_query_branching_dev(t._h, dq.data_ptr(), nq, bbits.data_ptr(), dcnt.data_ptr(), stream))
    torch.cuda.synchronize()
    succ, pred = dcnt >> 4, dcnt & 15
    assert int(succ.max()) <= 4 and int(pred.max()) <= 4
    assert bool((W.bits_to_bool(bbits, nq) == ((succ > 1) | (pred > 1))).all())
    b2 = torch.zeros_like(dbits)  # without counts the kernel may stop early: same bits
    L.check(t._lib.bft_gpu_query_branching_dev(t._h, dq.data_ptr(), nq, b2.data_ptr(), None, stream))
    torch.cuda.synchronize()
    assert torch.equal(b2, bbits)
    # counts against the membership of the eight neighbours, on a sample
    nc = 20000
    codes = S.unpack_codes(dq[:nc].cpu().numpy(), k)
    for side in (0, 1):
        tot = torch.zeros(nc, dtype=torch.int64, device=dev)
        for x in range(4):
            nb = np.concatenate([np.full((nc, 1), x, np.uint8), codes[:, :-1]], axis=1) if side == 0 else np.concatenate([codes[:, 1:], np.full((nc, 1), x, np.uint8)], axis=1)
            tot += W.member(allk, W.keys_of(torch.from_numpy(S.pack_codes(nb)).to(dev))).long()
        assert bool((tot == (pred if side == 0 else succ)[:nc].long()).all())
    # ---- colour rows (retrieveAnnotation) of 4x10^6 k-mers = 1 GB, device resident, against the inserting genomes ----
    nqc, rowbytes = 4_000_000, (G + 7) // 8
    drows = torch.zeros((nqc, rowbytes), dtype=torch.uint8, device=dev)
    dscr = torch.zeros(nqc, dtype=torch.int32, device=dev)
    L.check(t._lib.bft_gpu_query_color_rows_dev(t._h, dq.data_ptr(), nqc, dbits.data_ptr(), drows.data_ptr(), dscr.data_ptr(), stream))
    torch.cuda.synchronize()
    assert bool((W.bits_to_bool(dbits, nqc) == truth[:nqc]).all())
    assert bool(((drows != 0).any(dim=1) == truth[:nqc]).all())      # a row is non-zero exactly for the stored k-mers
    pop = torch.zeros(nqc, dtype=torch.int64, device=dev)
    for b in range(8):
        pop += ((drows >> b) & 1).sum(dim=1)
    ns = 50000                                                         # exact rows on a sample spread over the batch
    pick = torch.randperm(nqc, generator=g, device=dev)[:ns]
    exp = _colour_truth(torch, keys, qk[pick], W)
    sh = torch.arange(8, device=dev, dtype=torch.uint8)
    got = ((drows[pick][:, :, None] >> sh[None, None, :]) & 1).reshape(ns, rowbytes * 8)[:, :G].bool()
    assert bool((got == exp).all())
    assert bool((pop[pick] == exp.sum(dim=1)).all())
    # host entry points on a slice: id lists == rows
    nh = 20000
    hq = dq[:nh].cpu().numpy()
    hbits, off, ids = t.query_colors(hq)
    hrows = drows[:nh].cpu().numpy()
    unp = np.unpackbits(hrows, axis=1, bitorder="little")[:, :G]
    assert (np.diff(off.astype(np.int64)) == unp.sum(axis=1)).all()
    for i in range(0, nh, 211):
        assert ids[int(off[i]):int(off[i + 1])].tolist() == np.flatnonzero(unp[i]).tolist()
    t.close()


def test_sequence_queries_full_size(torch_dev):
    """SURVEY 8 f-4 at size: 10^6 reads x 150 nt on the 10-genome index, device-resident (bft_gpu_query_sequences_dev).  Ground truth
    without an oracle: per read and genome, the number of its k-mers that genome holds (set membership of every k-mer position) against
    ceil(positions x threshold); error-free reads must name their source genome at threshold 1, random reads nobody."""
    torch, dev = torch_dev
    import math
    from bloomfiltertrie_amd import BFT, workloads as W
    k, ngen, n_reads, rl = 27, 10, 1_000_000, 150
    pan = W.PanGenome(ngen, 2_000_000, 0.01, 4242, dev)
    t = BFT(k)
    keys, _ = W.build_index(t, pan, k)
    g = _gen(torch, dev, 5)
    src = torch.randint(0, ngen, (n_reads,), generator=g, device=dev)
    start = torch.randint(0, 2_000_000 - rl, (n_reads,), generator=g, device=dev)
    genomes = torch.stack([pan.genome(i) for i in range(ngen)])
    codes = genomes[src[:, None], start[:, None] + torch.arange(rl, device=dev)[None, :]]
    n_err = n_reads // 2  # the second half: 2 % substitutions; the last 5 %: random reads
    err = torch.rand((n_reads, rl), generator=g, device=dev) < 0.02
    err[:n_reads - n_err] = False
    codes = torch.where(err, (codes + torch.randint(1, 4, codes.shape, generator=g, device=dev, dtype=torch.uint8)) & 3, codes)
    n_rand = n_reads // 20
    codes[n_reads - n_rand:] = torch.randint(0, 4, (n_rand, rl), generator=g, device=dev, dtype=torch.uint8)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    blob = lut[codes.long()].contiguous().reshape(-1)
    off = (torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * rl).contiguous()
    rows = torch.zeros((n_reads, 2), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    m = rl - k + 1
    # ground truth on a slice: membership of every k-mer position in every genome's key table
    ns = 20_000
    sel = torch.cat([torch.arange(0, ns // 2, device=dev), torch.arange(n_reads - ns // 2, n_reads, device=dev)])
    win = codes[sel].unfold(1, k, 1).reshape(-1, k)  # [ns * m, k] codes, first nucleotide first
    sh = (2 * torch.arange(k, device=dev, dtype=torch.int64))[None, :]
    qk = (win.to(torch.int64) << sh).sum(dim=1)  # the packed layout: nucleotide j at bits 2j (src/fasta.c:11-23) = keys_of(pack_windows(...))
    cnt = torch.stack([W.member(keys[gi], qk).reshape(ns, m).sum(dim=1) for gi in range(ngen)], dim=1)  # [ns, ngen]
    for thr in (1.0, 0.8, 0.3):
        rows.fill_(0xFF)
        t.query_sequences_dev(blob.data_ptr(), off.data_ptr(), n_reads, n_reads * rl, thr, rows.data_ptr(), False, st)
        torch.cuda.synchronize()
        bits = torch.from_numpy(np.unpackbits(rows.cpu().numpy(), axis=1, bitorder="little")[:, :ngen].astype(bool)).to(dev)
        need = math.ceil(m * thr)
        assert bool((bits[sel] == ((cnt >= need) & (cnt > 0))).all()), thr
        if thr == 1.0:
            clean = n_reads - n_err
            assert bool(bits[torch.arange(clean, device=dev), src[:clean]].all())  # an error-free read names its source genome
        assert not bool(bits[n_reads - n_rand:].any())  # random reads: nobody holds 30 % of their k-mers
        assert not bool(np.unpackbits(rows.cpu().numpy(), axis=1, bitorder="little")[:, ngen:].any())  # padding bits stay zero
    # the host-buffer call gives the same rows
    some = [bytes(b) for b in blob.reshape(n_reads, rl)[:3000].cpu().numpy()]
    rows.fill_(0)
    t.query_sequences_dev(blob.data_ptr(), off.data_ptr(), n_reads, n_reads * rl, 0.8, rows.data_ptr(), False, st)
    torch.cuda.synchronize()
    dev_lists = [np.flatnonzero(r).tolist() for r in np.unpackbits(rows[:3000].cpu().numpy(), axis=1, bitorder="little")[:, :ngen]]
    assert t.query_sequences(some, 0.8) == dev_lists
