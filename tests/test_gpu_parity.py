"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs, bit-exact (integer/bit work: no tolerance), plus ground truth set semantics."""
import os

import numpy as np
import pytest

from bloomfiltertrie_amd import synth as S

pytestmark = pytest.mark.gpu


def _bft(k, **kw):
    from bloomfiltertrie_amd import BFT
    return BFT(k, device=0, **kw)


def _queries(km, k, seed=0, n_rand=None):
    rng = np.random.default_rng(seed)
    n_rand = n_rand or max(1000, len(km) // 2)
    parts = [km, S.pack_codes(rng.integers(0, 4, (n_rand, k), dtype=np.uint8))]
    if len(km):
        parts.append(S.snp_mutants(km, k, seed + 1))
    q = np.concatenate(parts)
    return np.ascontiguousarray(q[rng.permutation(len(q))])


def _check(t, o, km, k, seed=0):
    q = _queries(km, k, seed)
    got = t.query_presence(q)
    exp = o.query_presence(q)
    assert (got == exp).all()
    assert (S.from_bits(got, len(q)) == S.member(q, km)).all()
    return q


@pytest.mark.parametrize("k", [9, 18, 27, 36, 45, 63, 72, 99, 126])
def test_presence_random_genome(oracle_mod, k):
    km = S.distinct(S.kmers_of(S.random_genome(150000, 10 + k), k))
    t, o = _bft(k), oracle_mod.OracleBFT(k)
    t.insert_kmers(km, 0)
    o.insert_kmers(km, 0)
    _check(t, o, km, k)
    info = t.info()
    assert info["kmers"] == len(km) == o.stats()["kmers"]
    ek, _ = t.extract()
    assert sorted(S.row_keys(ek).tolist()) == sorted(S.row_keys(km).tolist())


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 254, 255, 256, 257, 300, 1000])
def test_tiny_and_ragged(oracle_mod, n):
    k = 27
    km = S.distinct(S.pack_codes(np.random.default_rng(n).integers(0, 4, (n, k), dtype=np.uint8))) if n else np.zeros((0, 7), np.uint8)
    t, o = _bft(k), oracle_mod.OracleBFT(k)
    t.insert_kmers(km, 0)
    o.insert_kmers(km, 0)
    t.build()
    for nq in (0, 1, 7, 8, 63, 64, 65, 255, 256, 257, 1023):
        q = _queries(km, k, seed=nq, n_rand=nq)[:nq] if nq else np.zeros((0, 7), np.uint8)
        got = t.query_presence(q)
        assert (got == o.query_presence(q)).all()


@pytest.mark.parametrize("k,levels", [(18, 1), (27, 1), (27, 2), (36, 2), (36, 3), (63, 3), (45, 4), (126, 5)])
def test_deep_tries(oracle_mod, k, levels):
    km = S.low_entropy_kmers(120000, k, 24, seed=k * 7 + levels, levels=levels)
    t, o = _bft(k), oracle_mod.OracleBFT(k)
    t.insert_kmers(km, 0)
    o.insert_kmers(km, 0)
    _check(t, o, km, k)
    assert t.info()["child_nodes"] > 0 and o.stats()["child_nodes"] > 0


def test_large_root_both_filter_geometries(oracle_mod):
    k = 27
    km = S.distinct(S.kmers_of(S.random_genome(1200000, 77), k))
    t, o = _bft(k), oracle_mod.OracleBFT(k)
    t.insert_kmers(km, 0)
    o.insert_kmers(km, 0)
    _check(t, o, km, k)
    info = t.info()
    assert 0 < info["ccs_s4"] < info["ccs"]


@pytest.mark.parametrize("k,ngen", [(27, 6), (18, 10), (36, 70), (27, 200)])
def test_colours(oracle_mod, k, ngen):
    anc = S.random_genome(3000 if ngen > 20 else 30000, 5)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 100 + g) if g else anc, k)) for g in range(ngen)]
    t, o = _bft(k), oracle_mod.OracleBFT(k)
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
        t.insert_kmers(km[:10], g)  # duplicate (k-mer, genome) pairs are no-ops, as in the reference
        o.insert_kmers(km, g)
    allk = S.distinct(np.concatenate(gk))
    q = _queries(allk, k, seed=3)
    bits, off, ids = t.query_colors(q)
    obits, ooff, oids = o.query_colors(q)
    assert (bits == obits).all() and (off == ooff).all() and (ids == oids).all()
    bits2, rows = t.query_color_rows(q)
    assert (bits2 == obits).all()
    pres = S.from_bits(obits, len(q))
    unp = np.unpackbits(rows, axis=1, bitorder="little")[:, :ngen]
    for i in np.flatnonzero(pres)[:3000]:
        assert np.flatnonzero(unp[i]).tolist() == oids[int(ooff[i]):int(ooff[i + 1])].tolist()
    assert not unp[~pres].any()
    # every stored k-mer and its colour set equals the oracle's extraction
    ek, ecs = t.extract()
    ok, ocs = o.extract()
    omap = {key: o.colorset(c) for key, c in zip(S.row_keys(ok).tolist(), ocs.tolist())}
    assert len(ek) == len(ok)
    cache = {}
    for key, c in zip(S.row_keys(ek).tolist(), ecs.tolist()):
        if c not in cache:
            cache[c] = t.colorset(c)
        assert cache[c] == omap[key]


@pytest.mark.parametrize("ngen,k", [(16, 27), (17, 27), (128, 27), (131, 27), (200, 27), (260, 27), (1999, 27), (131, 45), (520, 63), (9000, 27)])
def test_colour_rows_wide_against_ground_truth(ngen, k):
    """Fixed-width colour rows for row widths around the 16-byte chunks of k_color_rows_bm (2, 3, 16, 17, 25, 33, 250 bytes; 1125 bytes:
    a tile of 64 k-mers is more than 64 KiB): every byte of every row against the sets that were inserted; absent k-mers give all-zero
    rows.  The host call goes through row numbers and k_color_rows_bm16; the device-resident call looks the k-mers up inside the row kernel
    (k_color_rows_kh, rows of 16 bytes and up): both must give the same bytes."""
    import torch
    from bloomfiltertrie_amd import BFT
    km = S.distinct(S.kmers_of(S.random_genome(6000, 91), k))
    rng = np.random.default_rng(ngen)
    member = rng.random((ngen, len(km))) < (0.6 if ngen < 100 else 0.08)
    member[0, :] = True  # every k-mer is stored at least once
    t = BFT(k)
    for g in range(ngen):
        t.insert_kmers(np.ascontiguousarray(km[member[g]]), g)
    q = np.concatenate([km, S.snp_mutants(km[::3], k, 4)])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    bits, rows = t.query_color_rows(q)
    assert rows.shape == (len(q), (ngen + 7) // 8)
    pres = S.member(q, km)
    assert (S.from_bits(bits, len(q)).astype(bool) == pres).all()
    pos = {key: i for i, key in enumerate(S.row_keys(km).tolist())}
    exp = np.zeros((len(q), ngen), dtype=np.uint8)
    for i, key in enumerate(S.row_keys(q).tolist()):
        if key in pos:
            exp[i] = member[:, pos[key]]
    got = np.unpackbits(rows, axis=1, bitorder="little")
    assert (got[:, :ngen] == exp).all()
    assert not got[:, ngen:].any()  # padding bits of the last byte stay zero
    # odd batch sizes: the tail of the row stream is not a multiple of 16 bytes
    for m in (1, 5, 63, 1000):
        b2, r2 = t.query_color_rows(np.ascontiguousarray(q[:m]))
        assert (r2 == rows[:m]).all()
    # the device-resident call: lookup inside the row kernel; batch sizes around the tiles of 64 k-mers and the presence words
    dq = torch.from_numpy(q).to("cuda:0")
    for m in (len(q), 1, 63, 64, 65, 1000, 4097):
        m = min(m, len(q))
        dbits = torch.full((((m + 63) // 64) * 8,), 0xAA, dtype=torch.uint8, device="cuda:0")
        drows = torch.full((m, rows.shape[1]), 0x55, dtype=torch.uint8, device="cuda:0")
        scratch = torch.zeros(m, dtype=torch.int32, device="cuda:0")
        t.query_color_rows_dev(dq.data_ptr(), m, dbits.data_ptr(), drows.data_ptr(), scratch.data_ptr())
        torch.cuda.synchronize()
        assert (drows.cpu().numpy() == rows[:m]).all()
        assert (S.from_bits(dbits.cpu().numpy(), m).astype(bool) == pres[:m]).all()


def test_incremental_insert_and_rebuild(oracle_mod):
    k = 27
    anc = S.random_genome(50000, 9)
    t, o = _bft(k), oracle_mod.OracleBFT(k)
    allk = np.zeros((0, 7), np.uint8)
    for g in range(4):
        km = S.distinct(S.kmers_of(S.mutate(anc, 0.02, g), k))
        t.insert_kmers(km, g)
        o.insert_kmers(km, g)
        allk = S.distinct(np.concatenate([allk, km]))
        q = _queries(allk, k, seed=g)
        bits, off, ids = t.query_colors(q)  # lazily rebuilds the image
        obits, ooff, oids = o.query_colors(q)
        assert (bits == obits).all() and (off == ooff).all() and (ids == oids).all()
    assert t.info()["kmers"] == len(allk)


def test_device_resident_api_matches_host_api(oracle_mod):
    import torch
    k = 27
    km = S.distinct(S.kmers_of(S.random_genome(100000, 21), k))
    t = _bft(k)
    t.insert_kmers(km, 0)
    t.build()
    q = _queries(km, k, seed=5)
    dq = torch.from_numpy(q).cuda()
    dbits = torch.zeros(((len(q) + 63) // 64) * 8, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert t.kernel_time(reset=True) == (0.0, 0)  # timing is off until asked for: this call turns it on
    t.query_presence_dev(dq.data_ptr(), len(q), dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    got = dbits.cpu().numpy()[: (len(q) + 7) // 8]
    assert (got == t.query_presence(q)).all()
    ms, launches = t.kernel_time()
    assert launches >= 2 and ms > 0
    t.set_option("timing", 0)
    t.query_presence_dev(dq.data_ptr(), len(q), dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    with pytest.raises(Exception):
        t.set_option("debug_stop", 2)  # exists in the perf-probe build only (make probe), never in the shipped library


def test_rejects_bad_k():
    from bloomfiltertrie_amd import BFT
    from bloomfiltertrie_amd._lib import BFTError
    for k in (8, 127, 135):
        with pytest.raises(BFTError):
            BFT(k)


def test_out_of_order_genome_ids_ground_truth():
    """Genome ids arriving in decreasing order take the full (k-mer, genome) sort; colour sets stay sorted id lists."""
    k = 27
    anc = S.random_genome(20000, 8)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.02, 50 + g), k)) for g in range(5)]
    t = _bft(k)
    for g in (4, 2, 3, 0, 1):
        t.insert_kmers(gk[g], g)
    truth = {}
    for g, km in enumerate(gk):
        for key in S.row_keys(km).tolist():
            truth.setdefault(key, []).append(g)
    allk = S.distinct(np.concatenate(gk))
    q = _queries(allk, k, seed=9)
    bits, off, ids = t.query_colors(q)
    pres = S.from_bits(bits, len(q))
    for i, key in enumerate(S.row_keys(q).tolist()):
        exp = truth.get(key, [])
        assert pres[i] == bool(exp)
        assert ids[int(off[i]):int(off[i + 1])].tolist() == exp


@pytest.mark.parametrize("k,levels,ngen", [(27, 1, 6), (36, 3, 3), (63, 2, 70), (18, 1, 10)])
def test_bft_files_both_directions(oracle_mod, tmp_path, k, levels, ngen):
    """write_BFT / load_BFT: GPU image -> .bft -> restated reference reader+query; oracle .bft -> GPU image."""
    from bloomfiltertrie_amd import BFT
    base = S.low_entropy_kmers(60000, k, 24, seed=k + levels, levels=levels)
    gk = [base[:: (g % 4) + 1] for g in range(ngen)]
    t, o = _bft(k), oracle_mod.OracleBFT(k)
    for g, km in enumerate(gk):
        t.add_genome(f"g{g}.fa")
        t.insert_kmers(km, g)
        o.insert_kmers(km, g)
    q = _queries(base, k, seed=4)
    exp = o.query_colors(q)
    p1, p2 = str(tmp_path / "gpu.bft"), str(tmp_path / "orc.bft")
    t.write_bft(p1)
    o2 = oracle_mod.OracleBFT.load_bft(p1)
    assert o2.nb_genomes_loaded() == ngen
    got = o2.query_colors(q)
    assert all((a == b).all() for a, b in zip(got, exp))
    o.write_bft(p2, ngen)
    t2 = BFT.load_bft(p2)
    got2 = t2.query_colors(q)
    assert all((a == b).all() for a, b in zip(got2, exp))
    assert t2.info()["kmers"] == len(base) and t2.info()["genomes"] == ngen
    # a GPU-written file read back by the GPU path
    t3 = BFT.load_bft(p1)
    assert all((a == b).all() for a, b in zip(t3.query_colors(q), exp))


@pytest.mark.parametrize("k,deep", [(9, 0), (18, 0), (27, 0), (27, 2), (36, 0), (36, 3), (63, 0), (72, 0), (99, 2), (126, 0)])
def test_branching(oracle_mod, k, deep):
    if deep:
        km = S.low_entropy_kmers(40000, k, 16, seed=k, levels=deep)
    else:
        anc = S.random_genome(30000, 3)
        km = S.distinct(np.concatenate([S.kmers_of(g, k) for g in (anc, S.mutate(anc, 0.03, 1), S.mutate(anc, 0.03, 2))]))
    t, o = _bft(k), oracle_mod.OracleBFT(k)
    t.insert_kmers(km, 0)
    o.insert_kmers(km, 0)
    q = _queries(km, k, seed=2)
    obits, ocounts, onbr = o.query_branching(q)
    bits, counts = t.query_branching(q, with_counts=True)
    assert (counts == ocounts).all()
    assert (bits == obits).all()
    assert (t.query_branching(q) == obits).all()  # early-exit variant
    assert int(np.unpackbits(bits, bitorder="little")[: len(q)].sum()) == onbr


@pytest.mark.parametrize("k,load", [(27, 80), (63, 80), (99, 80), (27, 20), (126, 70)])
def test_branching_at_other_occupancies(oracle_mod, k, load):
    """The four successors (predecessors) of a k-mer share their home line in the k-mer hash and are counted by one masked scan: at 80 % the
    families run over several full lines (the scan goes on line by line, stops where the table's home order says so, and asks the overflow
    list after a run of full lines), at 20 % nearly every line has a free slot.  Counts and bits == the oracle's."""
    anc = S.random_genome(30000, 5)
    km = S.distinct(np.concatenate([S.kmers_of(g, k) for g in (anc, S.mutate(anc, 0.05, 1), S.mutate(anc, 0.05, 2), S.mutate(anc, 0.05, 3))]))
    t, o = _bft(k), oracle_mod.OracleBFT(k)
    t.set_option("kmer_hash_load", load)
    t.insert_kmers(km, 0)
    o.insert_kmers(km, 0)
    q = _queries(km, k, seed=4)
    obits, ocounts, onbr = o.query_branching(q)
    bits, counts = t.query_branching(q, with_counts=True)
    assert t.build_time()["kmer_hash_lines"] > 0
    assert (counts == ocounts).all() and (bits == obits).all()
    assert (t.query_branching(q) == obits).all()


@pytest.mark.parametrize("k,canonical", [(27, False), (27, True), (63, True), (18, False)])
def test_query_sequences(oracle_mod, k, canonical):
    ngen = 6
    anc = S.random_genome(8000, 21)
    strs = ["".join("ACGT"[c] for c in S.mutate(anc, 0.03, 60 + g)) for g in range(ngen)]
    comp = str.maketrans("ACGT", "TGCA")
    t, o = _bft(k), oracle_mod.OracleBFT(k)
    for g, s in enumerate(strs):
        kms = {s[i:i + k] for i in range(len(s) - k + 1)}
        if canonical:
            kms = {min(x, x[::-1].translate(comp)) for x in kms}
        packed, _ = S.ascii_to_packed(sorted(kms), k)
        t.insert_kmers(packed, g)
        o.insert_kmers(packed, g)
    rng = np.random.default_rng(5)
    reads = []
    for _ in range(300):
        g = int(rng.integers(0, ngen))
        a = int(rng.integers(0, len(strs[g]) - 400))
        r = strs[g][a:a + int(rng.integers(k - 3, 400))]
        if rng.random() < 0.5:
            r = r[::-1].translate(comp)
        if rng.random() < 0.2 and len(r) > 20:
            r = r[:15] + "N" + r[16:]
        reads.append(r)
    reads += ["", "ACGT", "acgt" * 20]
    for thr in (0.1, 0.75, 1.0):
        got = t.query_sequences(reads, thr, canonical)
        for r, gl in zip(reads, got):
            assert gl == o.query_sequence(r, thr, canonical, ngen), (r, thr)
    # the device-resident entry point: same rows, also from a blob that is not 16-byte aligned and without any padding behind it
    import torch
    dev = torch.device("cuda", 0)
    enc = [r.encode() for r in reads]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(e) for e in enc])
    blob = np.frombuffer(b"".join(enc), dtype=np.uint8)
    exp = t.query_sequences(reads, 0.75, canonical)
    for shift in (0, 3):
        d_blob = torch.zeros(len(blob) + shift, dtype=torch.uint8, device=dev)
        d_blob[shift:] = torch.from_numpy(blob.copy()).to(dev)
        d_off = torch.from_numpy(off).to(dev)
        d_rows = torch.full((len(enc), 1), 0xFF, dtype=torch.uint8, device=dev)
        t.query_sequences_dev(d_blob.data_ptr() + shift, d_off.data_ptr(), len(enc), len(blob), 0.75, d_rows.data_ptr(), canonical,
                              torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        unp = np.unpackbits(d_rows.cpu().numpy(), axis=1, bitorder="little")[:, :ngen]
        assert [np.flatnonzero(r).tolist() for r in unp] == exp, shift


def test_query_sequences_many_genomes_against_ground_truth():
    """Sequence queries with more genomes than one counter window of k_seq_tally holds (2048): every genome bit of every read
    against counts taken from the inserted sets, host call and device-resident call."""
    import math
    import torch
    from bloomfiltertrie_amd import BFT
    k, ngen, glen = 27, 2100, 3000
    g = S.random_genome(glen, 77)
    s = "".join("ACGT"[c] for c in g)
    packed, _ = S.ascii_to_packed([s[i:i + k] for i in range(glen - k + 1)], k)
    npos = glen - k + 1
    assert len(S.distinct(packed)) == npos
    rng = np.random.default_rng(3)
    member = rng.random((ngen, npos)) < 0.3
    member[5, :] = True
    member[2070, ::2] = True
    t = BFT(k)
    for gi in range(ngen):
        t.insert_kmers(np.ascontiguousarray(packed[member[gi]]), gi)
    reads, spans = [], []
    for _ in range(60):
        a = int(rng.integers(0, glen - 200))
        n = int(rng.integers(k, 200))
        reads.append(s[a:a + n])
        spans.append((a, n - k + 1))
    thr = 0.4
    got = t.query_sequences(reads, thr)
    for (a, m), gl in zip(spans, got):
        cnt = member[:, a:a + m].sum(axis=1)
        need = math.ceil(m * thr)
        assert gl == np.flatnonzero((cnt >= need) & (cnt > 0)).tolist()
    dev = torch.device("cuda", 0)
    enc = [r.encode() for r in reads]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(e) for e in enc])
    d_blob = torch.from_numpy(np.frombuffer(b"".join(enc), dtype=np.uint8).copy()).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_rows = torch.zeros((len(enc), (ngen + 7) // 8), dtype=torch.uint8, device=dev)
    t.query_sequences_dev(d_blob.data_ptr(), d_off.data_ptr(), len(enc), int(off[-1]), thr, d_rows.data_ptr(), False, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    unp = np.unpackbits(d_rows.cpu().numpy(), axis=1, bitorder="little")[:, :ngen]
    assert [np.flatnonzero(r).tolist() for r in unp] == got


def test_load_reference_shaped_file(oracle_mod, tmp_path):
    """load_BFT of a file with mode-3 annotations (comp_set_colors) and extended-annotation bytes."""
    from bloomfiltertrie_amd import BFT
    k, ngen = 27, 40
    base = S.low_entropy_kmers(50000, k, 24, seed=5, levels=2)
    rng = np.random.default_rng(7)
    o = oracle_mod.OracleBFT(k)
    for g in range(ngen):
        o.insert_kmers(np.ascontiguousarray(base[rng.random(len(base)) < rng.uniform(0.05, 0.9)]), g)
    q = _queries(base, k, seed=6)
    exp = o.query_colors(q)
    o.set_annotation_modes(comp=True, ext=True)
    p = str(tmp_path / "ref_shaped.bft")
    o.write_bft(p, ngen)
    t = BFT.load_bft(p)
    assert all((a == b).all() for a, b in zip(t.query_colors(q), exp))
    assert t.info()["genomes"] == ngen


# ---- extension beyond the reference: k % 9 != 0 (the headline metric names k = 31; the reference rejects it) ----
@pytest.mark.parametrize("k,deep", [(31, 0), (31, 2), (13, 0), (22, 1), (40, 3), (17, 0), (64, 0), (100, 2)])
def test_any_k_against_ground_truth(k, deep, tmp_path):
    from bloomfiltertrie_amd._lib import BFTError
    ngen = 4
    if deep:
        base = S.low_entropy_kmers(60000, k, 24, seed=k + deep, levels=deep)
    else:
        base = S.distinct(S.kmers_of(S.random_genome(80000, 3 + k), k))
    rng = np.random.default_rng(k)
    gk = [np.ascontiguousarray(base[rng.random(len(base)) < 0.6]) for _ in range(ngen)]
    t = _bft(k)
    truth = {}
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
        for key in map(bytes, km):
            truth.setdefault(key, []).append(g)
    q = _queries(base, k, seed=3)
    bits, off, ids = t.query_colors(q)
    pres = S.from_bits(bits, len(q))
    for i, key in enumerate(map(bytes, q)):
        exp = truth.get(key, [])
        assert pres[i] == bool(exp) and ids[int(off[i]):int(off[i + 1])].tolist() == exp
    assert (t.query_presence(q) == bits).all()
    # branching counts on a sample
    bb, bc = t.query_branching(q[:1500], with_counts=True)
    codes = S.unpack_codes(q[:1500], k)
    for i in range(0, 1500, 3):
        c = codes[i]
        succ = sum(bytes(S.pack_codes(np.concatenate([c[1:], [x]])[None, :])[0]) in truth for x in range(4))
        pred = sum(bytes(S.pack_codes(np.concatenate([[x], c[:-1]])[None, :])[0]) in truth for x in range(4))
        assert bc[i] == (succ << 4 | pred)
    ek, _ = t.extract()
    assert sorted(map(bytes, ek)) == sorted(truth)
    with pytest.raises(BFTError):  # the .bft format cannot express k % 9 != 0
        t.write_bft(str(tmp_path / "x.bft"))


def test_k31_dense_remainder_groups():
    rng = np.random.default_rng(0)
    base = rng.integers(0, 4, (30, 27), dtype=np.uint8)
    rem = np.array([[(i >> 6) & 3, (i >> 4) & 3, (i >> 2) & 3, i & 3] for i in range(256)], dtype=np.uint8)
    codes = np.concatenate([np.concatenate([np.repeat(base[j:j + 1], 256, 0), rem], axis=1) for j in range(30)])
    km = S.distinct(S.pack_codes(codes))
    t = _bft(31)
    t.insert_kmers(km, 0)
    q = np.concatenate([km, S.snp_mutants(km, 31, 1)])
    assert (S.from_bits(t.query_presence(q), len(q)) == S.member(q, km)).all()


@pytest.mark.parametrize("k,deep", [(27, False), (27, True), (63, True)])
def test_launch_options_do_not_change_answers(oracle_mod, k, deep):
    """Workgroup size, residency (k_query / k_query8 register budgets), grid multiplier and the suffix-group probe mode are tuning knobs only:
    presence, rows-based colours and branching are identical under every setting, and equal to the oracle's."""
    from bloomfiltertrie_amd import BFT
    km = S.low_entropy_kmers(120000, k, 12, seed=4, levels=2) if deep else S.distinct(S.kmers_of(S.random_genome(150000, 8), k))
    t, o = BFT(k), oracle_mod.OracleBFT(k)
    for g, part in enumerate(np.array_split(km, 3)):
        t.insert_kmers(np.ascontiguousarray(part), g)
        o.insert_kmers(np.ascontiguousarray(part), g)
    q = np.concatenate([km[::5], S.snp_mutants(km[::9], k, 2)])
    obits, ooff, oids = o.query_colors(q)
    ob, oc, _ = o.query_branching(q[:20000])
    t.build()
    tuned = t.build_time()["query_wgs_per_cu"]
    assert tuned in (1.0, 2.0, 3.0)
    assert t.build_time()["query_probe_rows"] in (4.0, 8.0)
    # (node prefix hash, residency, grid multiplier, probe rows, root tables, k-mer hash on/off, its occupancy, measured launch shape,
    #  the walk through the hash's regions, root quartile table, claimed chunks)
    for blk, wgs, mult, probe, rdir, kh, load, tune, wh, rq, dyn in [(1, 0, 1, 0, 1, 1, 50, 0, 0, 1, 1), (0, 1, 1, 4, 0, 0, 50, 0, 0, 0, 1), (2, 2, 1, 8, 2, 1, 80, 0, 1, 1, 0),
                                                                    (1, 3, 1, 4, 0, 0, 50, 1, 1, 0, 1), (0, 1, 3, 8, 3, 1, 10, 0, 1, 1, 1), (1, 2, 1, 0, 0, 0, 50, 0, 0, 1, 0),
                                                                    (0, 3, 2, 8, 1, 1, 65, 1, 0, 0, 1), (1, 3, 1, 0, 3, 0, 50, 0, 0, 1, 1), (1, 3, 1, 8, 2, 1, 60, 0, 1, 0, 1)]:
        t.set_option("kmer_hash_load", load)
        t.set_option("kmer_hash", kh)
        t.set_option("walk_hash", wh)
        t.set_option("root_quartiles", rq)
        t.set_option("query_dynamic", dyn)
        t.set_option("query_dynamic_min", 1024 if dyn else 1 << 16)
        t.set_option("node_hash", blk)
        t.set_option("query_wgs_per_cu", wgs)
        t.set_option("query_grid_mult", mult)
        t.set_option("query_probe", probe)
        t.set_option("root_direct", rdir)
        t.set_option("tune", tune)
        assert (t.build_time()["kmer_hash_lines"] > 0) == bool(kh)
        bits, off, ids = t.query_colors(q)
        assert (bits == obits).all() and (off == ooff).all() and (ids == oids).all(), (blk, wgs, mult, probe, rdir, kh, load)
        assert (t.query_presence(q) == obits).all()
        bb, bc = t.query_branching(q[:20000], with_counts=True)
        assert (bb == ob).all() and (bc == oc).all(), (blk, wgs, mult, probe, rdir, kh, load)
        assert (t.query_branching(q[:20000]) == ob).all()
    with pytest.raises(Exception):
        t.set_option("query_block", 1024)  # (a round-1 knob: gone)
    with pytest.raises(Exception):
        t.set_option("query_wgs_per_cu", 4)
    with pytest.raises(Exception):
        t.set_option("query_probe", 16)
    with pytest.raises(Exception):
        t.set_option("query_bucket_bits", 8)  # (round 2's prefix-bucketed batches: gone)
    with pytest.raises(Exception):
        t.set_option("kmer_hash_load", 95)


@pytest.mark.parametrize("k,levels", [(27, 0), (27, 2), (31, 0), (32, 0), (36, 0), (63, 2), (72, 0), (126, 0)])
def test_kmer_hash_answers_like_the_container_walk(oracle_mod, k, levels):
    """The k-mer hash (BFT_KH_*: every stored k-mer in one table of 64-byte lines) gives bit-identical presence bitmaps, colour sets
    and colour rows to the container walk -- over the sorted table ("kmer_hash" 0) and with the plain root groups looked up in their regions
    of the table ("walk_hash" 1) -- ragged batch sizes (not a multiple of 64), all-absent batches, k-mers that share their first 8
    nucleotides, every occupancy, every key width (k = 27 ... 126: one to four words, 8 to 1 slots per line); and all of them agree with
    ground truth and with the oracle where it exists (k % 9 == 0)."""
    from bloomfiltertrie_amd import BFT
    anc = S.random_genome(150000, 3 + k)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.02, 50 + g), k)) for g in range(3)]
    if levels:
        gk[0] = np.concatenate([gk[0], S.low_entropy_kmers(60000, k, 12, seed=9, levels=levels)])
    t = BFT(k)
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
    t.build()
    assert t.build_time()["kmer_hash_lines"] > 0 and t.footprint()["kmer_hash"] > 0  # every k has the table
    allk = S.distinct(np.concatenate(gk))
    rng = np.random.default_rng(k)
    base = np.concatenate([allk, S.snp_mutants(allk, k, 1), S.pack_codes(rng.integers(0, 4, (5000, k), dtype=np.uint8))])
    one_prefix = base[(base[:, 0] == base[0, 0]) & (base[:, 1] == base[0, 1])]  # same first 8 nucleotides
    absent = base[~S.member(base, allk).astype(bool)]
    for n, src in ((300_007, base), (64, base), (1, base), (8192, base), (8193, base), (len(one_prefix), one_prefix), (5001, absent)):
        q = np.ascontiguousarray(src[rng.integers(0, len(src), n)])
        t.set_option("kmer_hash", 0)
        ref_bits, ref_rows, ref_sets = t.query_rows(q)
        ref_b3, ref_crows = t.query_color_rows(q)
        truth = S.member(q, allk)
        assert (S.from_bits(ref_bits, n).astype(bool) == truth).all() and (ref_b3 == ref_bits).all()
        for load in (60, 80, 20):
            t.set_option("kmer_hash_load", load)
            t.set_option("kmer_hash", 1)
            assert (t.query_presence(q) == ref_bits).all(), (k, n, load)
            t.set_option("walk_hash", 1)
            assert (t.query_presence(q) == ref_bits).all(), (k, n, load, "walk_hash")
            bw, crw = t.query_color_rows(q)
            assert (bw == ref_bits).all() and (crw == ref_crows).all(), (k, n, load, "walk_hash")
            t.set_option("walk_hash", 0)
            b2, r2, s2 = t.query_rows(q)
            assert (b2 == ref_bits).all() and (r2 == ref_rows).all() and (s2 == ref_sets).all(), (k, n, load)
            b3, crows = t.query_color_rows(q)
            assert (b3 == ref_bits).all() and (crows == ref_crows).all(), (k, n, load)
            b4, off4, ids4 = t.query_colors(q[:20000])
            sizes = np.diff(off4)
            assert (b4 == ref_bits[: (min(n, 20000) + 7) // 8]).all()
            assert ((sizes > 0) == truth[:20000].astype(bool)).all()
    if k % 9 == 0:
        o = oracle_mod.OracleBFT(k)
        for g, km in enumerate(gk):
            o.insert_kmers(np.ascontiguousarray(km), g)
        q = np.ascontiguousarray(base[rng.integers(0, len(base), 50_000)])
        bits, off, ids = t.query_colors(q)
        obits, ooff, oids = o.query_colors(q)
        assert (bits == obits).all() and (off == ooff).all() and (ids == oids).all()
        bb, bc = t.query_branching(q[:20000], with_counts=True)
        ob, oc, _ = o.query_branching(q[:20000])
        assert (bb == ob).all() and (bc == oc).all()
    t.close()


@pytest.mark.parametrize("k", [18, 27, 31, 32, 45, 63, 64, 90, 126])
def test_kmer_hash_table_invariants(k):
    """The table the GPU builds holds every stored k-mer exactly once with its colour set as the value, nothing else, and every line between
    a k-mer's home line and its own is full -- the property the lookup's early exit relies on (every slot decoded on the host and checked
    against the sorted table: bft_hosttest_kh_verify).  And the layout is CANONICAL -- a function of the stored set alone --: the GPU build (a
    device-wide sort by home line with the k-mers as payload, a max-scan, every line assembled and stored once) gives, byte for byte, the table
    and the overflow list of the sequential host restatement (bft_kh_host.h; tests/test_abi_and_host.py checks the restatement itself:
    lookups, decoding of every slot, occupancy)."""
    import ctypes as C
    from bloomfiltertrie_amd import BFT, _lib
    W = (2 * k + 63) // 64
    anc = S.random_genome(200000, k)
    hostlib = C.CDLL(os.path.join(_lib.CSRC, "libbft_hosttest.so"))
    hostlib.bft_hosttest_kh_build.restype = C.c_uint64
    hostlib.bft_hosttest_kh_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    hostlib.bft_hosttest_kh_verify.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                               C.c_uint32]
    for load in (55, 80):  # (80 %: runs of full lines beyond the displacement bits -- the overflow list)
        canonical = 1
        t = BFT(k)
        t.set_option("kmer_hash_load", load)
        t.set_option("compact_table", 0)
        for g in range(3):
            km = S.distinct(S.kmers_of(S.mutate(anc, 0.02, g), k))
            if g == 0:
                km = np.concatenate([km, S.low_entropy_kmers(30000, k, 6, seed=k, levels=1)])  # thousands of k-mers under a handful of root prefixes
            t.insert_kmers(km, g)
        t.build()
        kh = np.ascontiguousarray(t.debug_array("kh", np.uint64))
        tk = np.ascontiguousarray(t.debug_array("tk", np.uint64).reshape(-1, W))
        tcol = np.ascontiguousarray(t.debug_array("tcol", np.uint32))
        ovk = np.ascontiguousarray(t.debug_array("kh_ovf_k", np.uint64))
        ovv = np.ascontiguousarray(t.debug_array("kh_ovf_v", np.uint32))
        n_sets = t.info()["colorsets"]
        bt = t.build_time()
        home_lines, db, maxd, novf = int(bt["kmer_hash_lines"]), int(bt["kmer_hash_dbits"]), int(bt["kmer_hash_maxd"]), int(bt["kmer_hash_overflow"])
        assert home_lines > 0 and len(kh) == (home_lines + 256) * 8 and maxd < (1 << db) and len(ovv) == novf and len(ovk) == novf * W
        rc = hostlib.bft_hosttest_kh_verify(tk.ctypes.data, tcol.ctypes.data, len(tk), k, n_sets, load, maxd, kh.ctypes.data, len(kh) // 8, ovk.ctypes.data, ovv.ctypes.data, novf)
        assert rc == 1, (rc, canonical, load)
        if canonical:
            lines = np.zeros(len(kh) + 4096, np.uint64)
            geo = np.zeros(14, np.uint32)
            hk, hv = np.zeros(4096 * W, np.uint64), np.zeros(4096, np.uint32)
            nw = hostlib.bft_hosttest_kh_build(tk.ctypes.data, tcol.ctypes.data, len(tk), k, n_sets, load, lines.ctypes.data, len(lines), geo.ctypes.data, hk.ctypes.data, hv.ctypes.data)
            assert nw == len(kh) and int(geo[10]) == home_lines and int(geo[12]) == db and int(geo[13]) == novf and int(geo[11]) == maxd, (nw, len(kh), geo)
            assert (kh == lines[:nw]).all() and (ovk == hk[: novf * W]).all() and (ovv == hv[:novf]).all()
        # every stored k-mer is found with its colour set, whichever way it got into the table (or its overflow list)
        km_all, cs_all = t.extract()
        assert S.from_bits(t.query_presence(km_all), len(km_all)).all()
        b1, cr1 = t.query_color_rows(km_all[:100000])   # (device path: colour sets out of the table's slots)
        t.set_option("kmer_hash", 0)
        b2, cr2 = t.query_color_rows(km_all[:100000])   # (the walk: rows -> colour sets)
        assert (b1 == b2).all() and (cr1 == cr2).all()
        t.close()


@pytest.mark.parametrize("k,per_prefix", [(18, 20), (27, 12), (27, 120), (36, 30), (63, 200), (31, 40)])
def test_suffix_groups_and_root_tables_against_oracle(oracle_mod, k, per_prefix):
    """Suffix groups of a dozen to 200 rows (block probes of the sorted table) and the root level through the derived range /
    direct tables: presence, rows and colour sets equal the oracle's (k % 9 == 0) and ground truth, with each accelerator on
    and off, with and without the k-mer hash in front."""
    from bloomfiltertrie_amd import BFT
    n_pref = 1500
    km = S.low_entropy_kmers(n_pref * per_prefix, k, n_pref, seed=k + per_prefix, levels=1)
    rng = np.random.default_rng(5)
    parts = np.array_split(km[rng.permutation(len(km))], 3)
    t = BFT(k)
    o = k % 9 == 0
    for g, part in enumerate(parts):
        t.insert_kmers(np.ascontiguousarray(part), g)
        t.insert_kmers(np.ascontiguousarray(part[::7]), (g + 1) % 3)
    t.build()
    q = np.concatenate([km, S.snp_mutants(km, k, 3), S.snp_mutants(km[::2], k, 4), S.pack_codes(rng.integers(0, 4, (20000, k), dtype=np.uint8))])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    truth = S.member(q, km)
    ref = None
    for gh, rd in ((1, 2), (0, 2), (1, 1), (1, 0), (0, 0)):
        t.set_option("kmer_hash", gh)
        t.set_option("root_direct", rd)
        bits, rows, sets = t.query_rows(q)
        assert (S.from_bits(bits, len(q)).astype(bool) == truth).all(), (gh, rd)
        if ref is None:
            ref = (bits, rows, sets)
            stored, cs = t.extract()
            assert (stored[rows[truth]] == q[truth]).all() and (cs[rows[truth]] == sets[truth]).all()
        else:
            assert (bits == ref[0]).all() and (rows == ref[1]).all() and (sets == ref[2]).all(), (gh, rd)
    t.set_option("kmer_hash", 1)
    t.set_option("root_direct", 2)
    if o:
        o2 = oracle_mod.OracleBFT(k)
        for g, part in enumerate(parts):
            both = np.concatenate([part, parts[(g + 2) % 3][::7]])  # genome g also holds every 7th k-mer of part g-1 (as inserted above)
            o2.insert_kmers(np.ascontiguousarray(both), g)
        bits, off, ids = t.query_colors(q[:60000])
        obits, ooff, oids = o2.query_colors(q[:60000])
        assert (bits == obits).all() and (off == ooff).all() and (ids == oids).all()
        bb, bc = t.query_branching(q[:20000], with_counts=True)
        ob, oc, _ = o2.query_branching(q[:20000])
        assert (bb == ob).all() and (bc == oc).all()
    t.close()


def test_queries_on_two_caller_streams_around_a_rebuild():
    """*_dev calls on two different caller streams, nothing synchronised by the caller, then an insertion + rebuild (which releases the
    image arrays the queries read) and set_option calls that re-derive tables: every bitmap still equals ground truth."""
    import torch
    from bloomfiltertrie_amd import BFT
    k = 27
    km = S.distinct(S.kmers_of(S.random_genome(400000, 31), k))
    half = len(km) // 2
    t = BFT(k)
    t.insert_kmers(np.ascontiguousarray(km[:half]), 0)
    t.build()
    dev = torch.device("cuda", 0)
    q = np.ascontiguousarray(np.concatenate([km, S.snp_mutants(km[::4], k, 9)]))
    dq = torch.from_numpy(q).to(dev)
    n = len(q)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    outs = [torch.zeros(((n + 63) // 64) * 8, dtype=torch.uint8, device=dev) for _ in range(6)]
    torch.cuda.synchronize()
    for i in range(4):
        t.query_presence_dev(dq.data_ptr(), n, outs[i].data_ptr(), (s1 if i % 2 == 0 else s2).cuda_stream)
    t.insert_kmers(np.ascontiguousarray(km[half:]), 1)  # the next query rebuilds
    t.query_presence_dev(dq.data_ptr(), n, outs[4].data_ptr(), s2.cuda_stream)
    t.set_option("node_hash", 0)
    t.set_option("root_direct", 1)
    t.query_presence_dev(dq.data_ptr(), n, outs[5].data_ptr(), s1.cuda_stream)
    torch.cuda.synchronize()
    first = S.member(q, S.distinct(km[:half]))
    both = S.member(q, km)
    for i in range(4):
        assert (S.from_bits(outs[i].cpu().numpy(), n).astype(bool) == first).all(), i
    for i in (4, 5):
        assert (S.from_bits(outs[i].cpu().numpy(), n).astype(bool) == both).all(), i


@pytest.mark.parametrize("k,devices", [(27, [0]), (27, [0, 0]), (63, [0, 0, 0]), (31, [0, 0])])
def test_device_group_answers_like_one_handle(k, devices):
    """bft_gpu_group_*: the index replicated per device slot (the blob packed, copied, unpacked; the same device may serve several slots,
    which is how a one-GPU box exercises replication and sharding), a host batch cut into 64-aligned slices, one persistent host thread per slot
    that moves its slice through its own pinned staging slots:
    presence, colour rows and branching equal the single handle's, for ragged sizes around the slice boundaries."""
    from bloomfiltertrie_amd import BFT, BFTGroup
    anc = S.random_genome(120000, k)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.02, 30 + g), k)) for g in range(5)]
    t = BFT(k)
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
    grp = BFTGroup(t, devices)
    assert grp.size() == len(devices)
    # "compact_table" (the default) holds for every member, the source included: the sorted table that travelled in the blob, and the one the
    # source brought back to pack it, are not resident once the group exists (small host batches and rows bring a member's table back later)
    for i in range(grp.size()):
        fp = grp.member_footprint(i)
        assert fp["kmer_hash"] > 0 and fp["kmer_table"] <= 8 and fp["colorset_per_kmer"] <= 8, (i, fp)
    assert t.footprint()["kmer_table"] <= 8
    allk = S.distinct(np.concatenate(gk))
    rng = np.random.default_rng(k)
    base = np.concatenate([allk, S.snp_mutants(allk, k, 2)])
    for n in (1, 63, 64, 65, 129, 100_003, 400_000):
        q = np.ascontiguousarray(base[rng.integers(0, len(base), n)])
        assert (grp.query_presence(q) == t.query_presence(q)).all(), n
        gb, grows = grp.query_color_rows(q)
        tb, trows = t.query_color_rows(q)
        assert (gb == tb).all() and (grows == trows).all(), n
        b1, c1 = grp.query_branching(q[:50_000], with_counts=True)
        b2, c2 = t.query_branching(q[:50_000], with_counts=True)
        assert (b1 == b2).all() and (c1 == c2).all(), n
    assert (S.from_bits(grp.query_presence(base), len(base)) == S.member(base, allk)).all()
    if len(devices) == 2 and k == 27:
        # slices of more than one staging chunk (2^22 k-mers): both pinned slots of a member in use, chunks retired out of step with the copies in
        n = 9_000_037
        q = np.ascontiguousarray(base[rng.integers(0, len(base), n)])
        assert (grp.query_presence(q) == t.query_presence(q)).all()
        gb, grows = grp.query_color_rows(q)
        tb, trows = t.query_color_rows(q)
        assert (gb == tb).all() and (grows == trows).all()
        b1, c1 = grp.query_branching(q, with_counts=True)
        b2, c2 = t.query_branching(q, with_counts=True)
        assert (b1 == b2).all() and (c1 == c2).all()
        assert (grp.query_branching(q) == b2).all()
    grp.close()
    with pytest.raises(Exception):
        BFTGroup(t, [99])
    t.close()


@pytest.mark.parametrize("k", [27, 63])
def test_device_group_on_resident_batches(k):
    """bft_gpu_group_*_dev: every slot answers the shard that lies in its GPU's memory on a stream of its own; nothing is synchronised by the
    calls (the slots run side by side) -- presence, colour rows and branching of the shards equal the single handle's on the whole batch.
    Three slots on the one device of this box: the shards of bft_gpu_group_shard."""
    import torch
    from bloomfiltertrie_amd import BFT, BFTGroup, shard
    anc = S.random_genome(120000, k + 1)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.02, 40 + g), k)) for g in range(9)]
    t = BFT(k)
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
    grp = BFTGroup(t, [0, 0, 0])
    assert [grp.member_device(i) for i in range(3)] == [0, 0, 0] and grp.member_device(3) == -1
    with pytest.raises(ValueError):  # (one entry per slot: the C side indexes bft_gpu_group_size entries of every array)
        grp.query_presence_dev([0, 0], [0, 0, 0], [0, 0, 0])
    with pytest.raises(Exception):  # (a slot with k-mers and no buffers is refused before anything is enqueued)
        grp.query_presence_dev([0, 0, 0], [64, 0, 0], [0, 0, 0])
    allk = S.distinct(np.concatenate(gk))
    rng = np.random.default_rng(k)
    base = np.concatenate([allk, S.snp_mutants(allk, k, 2)])
    n = 300_007
    q = np.ascontiguousarray(base[rng.integers(0, len(base), n)])
    ref_bits = t.query_presence(q)
    ref_b2, ref_rows = t.query_color_rows(q)
    ref_br, ref_cnt = t.query_branching(q, with_counts=True)
    rowbytes = ref_rows.shape[1]
    dev = torch.device("cuda", 0)
    parts = [shard(n, 3, i) for i in range(3)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    dq = [torch.from_numpy(q[a:b]).to(dev) for a, b in parts]
    ns = [b - a for a, b in parts]
    bits = [torch.zeros(((m + 63) // 64) * 8, dtype=torch.uint8, device=dev) for m in ns]
    rows = [torch.zeros((m, rowbytes), dtype=torch.uint8, device=dev) for m in ns]
    scratch = [torch.zeros(m, dtype=torch.int32, device=dev) for m in ns]
    brb = [torch.zeros(((m + 63) // 64) * 8, dtype=torch.uint8, device=dev) for m in ns]
    cnt = [torch.zeros(m, dtype=torch.uint8, device=dev) for m in ns]
    torch.cuda.synchronize()
    sp = [s.cuda_stream for s in streams]
    grp.query_presence_dev([x.data_ptr() for x in dq], ns, [x.data_ptr() for x in bits], sp)
    torch.cuda.synchronize()
    got = np.concatenate([x.cpu().numpy()[: (m + 7) // 8] for x, m in zip(bits, ns)])  # (shard starts are multiples of 64: byte ranges of the bitmap)
    assert (got[: (n + 7) // 8] == ref_bits).all()
    grp.query_color_rows_dev([x.data_ptr() for x in dq], ns, [x.data_ptr() for x in bits], [x.data_ptr() for x in rows], [x.data_ptr() for x in scratch], sp)
    grp.query_branching_dev([x.data_ptr() for x in dq], ns, [x.data_ptr() for x in brb], [x.data_ptr() for x in cnt], sp)
    torch.cuda.synchronize()
    assert (np.concatenate([x.cpu().numpy() for x in rows]) == ref_rows).all()
    assert (np.concatenate([x.cpu().numpy()[: (m + 7) // 8] for x, m in zip(brb, ns)])[: (n + 7) // 8] == ref_br).all()
    assert (np.concatenate([x.cpu().numpy() for x in cnt]) == ref_cnt).all()
    # a slot without work, the slots' own streams
    grp.query_presence_dev([dq[0].data_ptr(), 0, dq[2].data_ptr()], [ns[0], 0, ns[2]], [bits[0].data_ptr(), 0, bits[2].data_ptr()], None)
    torch.cuda.synchronize()
    grp.close()
    t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [27, 63])
def test_claim_counters_survive_an_unfinished_launch(k):
    """The query kernels claim their blocks from a per-stream counter (bft_claims.h).  The counter only grows and every launch raises it to the start
    of its own range, so a launch that never finished -- its counter left anywhere inside the range it was given -- cannot make the next one skip
    blocks: "test_stale_claims" leaves every counter one below the next base / exactly there / back at zero, and presence (k-mer hash, container walk
    with and without hashed root groups), branching and sequence queries still answer whole batches, bit for bit as before."""
    import torch
    from bloomfiltertrie_amd import BFT
    anc = S.random_genome(150000, 3 * k)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.02, 70 + g), k)) for g in range(4)]
    t = BFT(k)
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
    allk = S.distinct(np.concatenate(gk))
    rng = np.random.default_rng(k)
    base = np.concatenate([allk, S.snp_mutants(allk, k, 2)])
    q = np.ascontiguousarray(base[rng.integers(0, len(base), 3_000_000)])  # (claims start at 2^16 k-mers; the resident grid covers 2^21 in its first round)
    truth = S.member(q, allk)
    dq = torch.from_numpy(q).cuda()
    nq = len(q)
    stream = torch.cuda.current_stream().cuda_stream

    def presence():
        bits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device="cuda")
        t.query_presence_dev(dq.data_ptr(), nq, bits.data_ptr(), stream)
        torch.cuda.synchronize()
        return bits.cpu().numpy()[: (nq + 7) // 8]

    ref = presence()
    assert (S.from_bits(ref, nq) == truth).all()
    br_ref = t.query_branching(q[:1_000_000], with_counts=True)
    for mode in (1, 2, 3, 1):
        for opts in ((), (("walk_hash", 1),), (("kmer_hash", 0),)):
            for nm, v in opts:
                t.set_option(nm, v)
            presence()  # (a launch on this form first, so that its stream slot exists and has moved on)
            t.set_option("test_stale_claims", mode)
            assert (presence() == ref).all(), (mode, opts)
            assert (presence() == ref).all(), (mode, opts)  # ... and the one after it
            for nm, v in opts:
                t.set_option(nm, {"walk_hash": 0, "kmer_hash": 1}[nm])
        t.set_option("test_stale_claims", mode)
        b, c = t.query_branching(q[:1_000_000], with_counts=True)
        assert (b == br_ref[0]).all() and (c == br_ref[1]).all(), mode
    t.close()


@pytest.mark.gpu
def _hip_runtime():
    """the HIP runtime this process already has loaded (torch's): raw streams beyond torch's pool of 32"""
    import ctypes
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("no libamdhip64 mapped")


def test_more_streams_than_claim_slots():
    """A handle keeps a claim counter for 32 streams; a further stream takes over the least recently used slot once that slot's last launch is over
    (rounds 3-4 sent every launch beyond the 32nd stream down the static split for the life of the handle).  Streams made with hipStreamCreate
    (torch.cuda.Stream() hands out a pool of 32 again and again): 40 streams used once and DESTROYED -- a request-per-stream server --, then 40 new
    ones, each queried twice: same bits everywhere, and not one launch that ran static."""
    import ctypes
    import torch
    from bloomfiltertrie_amd import BFT
    hip = _hip_runtime()
    k = 27
    anc = S.random_genome(150000, 9)
    km = S.distinct(S.kmers_of(anc, k))
    t = BFT(k)
    t.insert_kmers(km, 0)
    rng = np.random.default_rng(1)
    base = np.concatenate([km, S.snp_mutants(km, k, 2)])
    q = np.ascontiguousarray(base[rng.integers(0, len(base), 3_000_000)])
    dq = torch.from_numpy(q).cuda()
    nq = len(q)
    ref = None
    seen = set()

    def new_stream():
        st = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(st)) == 0
        return st

    def ask(st):
        nonlocal ref
        bits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        t.query_presence_dev(dq.data_ptr(), nq, bits.data_ptr(), st.value)
        assert hip.hipStreamSynchronize(st) == 0
        got = bits.cpu().numpy()
        if ref is None:
            ref = got
            assert (S.from_bits(ref[: (nq + 7) // 8], nq) == S.member(q, km)).all()
        assert (got == ref).all()

    for _ in range(40):  # one request, one stream
        st = new_stream()
        seen.add(st.value)
        ask(st)
        assert hip.hipStreamDestroy(st) == 0
    streams = [new_stream() for _ in range(40)]
    for rnd in range(2):
        for st in streams:
            seen.add(st.value)
            ask(st)
    for st in streams:
        assert hip.hipStreamDestroy(st) == 0
    assert len(seen) > 32  # (more distinct streams than slots: the slots did change hands)
    assert t.build_time()["claims_static_launches"] == 0
    t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k,ngen", [(27, 12), (63, 300), (18, 3)])
def test_resident_id_lists_through_the_kmer_hash(k, ngen):
    """bft_gpu_query_colors_dev: get_annotation + get_list_id_genomes (src/bft.c:363-387, 622-641) on a resident batch -- offsets and ids in HBM equal
    the oracle's lists; the sorted table is NOT brought back ("compact_table": the hash line holds the colour set), a buffer that is too small gets
    the count and never more than its capacity, and the same call answers through the container walk (kmer_hash 0)."""
    import torch
    from bloomfiltertrie_amd import BFT
    from oracle import oracle as O
    anc = S.random_genome(40000, k + ngen)
    t, o = BFT(k), O.OracleBFT(k)
    rng = np.random.default_rng(k)
    base = S.distinct(S.kmers_of(anc, k))
    for g in range(ngen):
        km = np.ascontiguousarray(base[rng.random(len(base)) < (0.9 if g % 7 == 0 else 0.05)])
        t.insert_kmers(km, g)
        o.insert_kmers(km, g)
    t.build()
    allk, _ = o.extract()
    q = np.concatenate([allk[::3], S.snp_mutants(allk[::5], k, 1)])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    obits, ooff, oids = o.query_colors(q)
    n = len(q)
    dq = torch.from_numpy(q).cuda()
    for form in ("kmer_hash", "walk"):
        if form == "walk":
            t.set_option("kmer_hash", 0)
        else:
            assert t.footprint()["kmer_table"] <= 8
        bits = torch.zeros(((n + 63) // 64) * 8, dtype=torch.uint8, device="cuda")
        off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        need = torch.zeros(1, dtype=torch.int64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        small = torch.full((16,), -1, dtype=torch.int32, device="cuda")
        t.query_colors_dev(dq.data_ptr(), n, bits.data_ptr(), off.data_ptr(), small.data_ptr(), 16, need.data_ptr(), stream)  # too small: only the count
        torch.cuda.synchronize()
        total = int(need.item())
        assert total == len(oids) and total > 16
        # (too small: through the k-mer hash -- one launch, which knows the total at its end -- the first 16 ids and not a byte more; through the walk nothing)
        assert (small.cpu().numpy().astype(np.uint32) == oids[:16]).all() if form == "kmer_hash" else bool((small == -1).all())
        ids = torch.zeros(total, dtype=torch.int32, device="cuda")
        t.query_colors_dev(dq.data_ptr(), n, bits.data_ptr(), off.data_ptr(), ids.data_ptr(), total, need.data_ptr(), stream)
        torch.cuda.synchronize()
        assert (bits.cpu().numpy()[: (n + 7) // 8] == obits).all(), form
        assert (off.cpu().numpy().astype(np.uint64) == ooff).all(), form
        assert (ids.cpu().numpy().astype(np.uint32) == oids).all(), form
        if form == "kmer_hash":
            assert t.footprint()["kmer_table"] <= 8  # (the resident call never needs the sorted table)
            hb, hoff, hids = t.query_colors(q)  # the host entry point is the same path on staged chunks
            assert (hb == obits).all() and (hoff == ooff).all() and (hids == oids).all()
            assert t.footprint()["kmer_table"] <= 8
    t.close()


@pytest.mark.gpu
def test_captured_queries_replay():
    """A *_dev presence query recorded into a HIP graph and replayed: every replay answers every block (claim counters hand a launch the start of its
    range as a kernel argument, which a replay cannot move: a launch on a stream that is being captured runs static rounds), direct launches on the same
    stream in between do not disturb it, and the batch is large enough for the claimed rounds to be on (2^16 k-mers and more)."""
    import torch
    from bloomfiltertrie_amd import BFT
    k = 27
    km = S.distinct(S.kmers_of(S.random_genome(300000, 23), k))
    rng = np.random.default_rng(5)
    q = np.ascontiguousarray(np.concatenate([km[rng.integers(0, len(km), 150000)], S.snp_mutants(km[rng.integers(0, len(km), 150000)], k, 9)]))
    want = S.member(q, km)
    with BFT(k) as t:
        t.insert_kmers(km, 0)
        t.build()
        dev = torch.device("cuda", 0)
        dq = torch.from_numpy(q).to(dev)
        bits = torch.zeros((len(q) + 63) // 64 * 8, dtype=torch.uint8, device=dev)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            t.query_presence_dev(dq.data_ptr(), len(q), bits.data_ptr(), s.cuda_stream)  # (a direct launch first: the stream's counter has moved)
        s.synchronize()
        assert (S.from_bits(bits.cpu().numpy(), len(q)) == want).all()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            t.query_presence_dev(dq.data_ptr(), len(q), bits.data_ptr(), torch.cuda.current_stream().cuda_stream)
        for rep in range(3):
            bits.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert (S.from_bits(bits.cpu().numpy(), len(q)) == want).all(), rep
            with torch.cuda.stream(s):  # a direct launch between the replays
                t.query_presence_dev(dq.data_ptr(), len(q), bits.data_ptr(), s.cuda_stream)
            s.synchronize()
            assert (S.from_bits(bits.cpu().numpy(), len(q)) == want).all()


def test_captured_colour_queries_replay():
    """The one-launch colour queries of a resident batch (id lists: k_colors_kh, whose tile counter and look-back states are zeroed by a memset
    that is captured with it; bitmap rows: k_color_rows_kh) recorded into HIP graphs and replayed with other k-mers in the same buffers: every
    replay equals the direct call's answers.  (The handle's scratch is sized by a direct call first: nothing allocates while a stream is captured.)"""
    import torch
    from bloomfiltertrie_amd import BFT
    k, ngen = 27, 140  # (18-byte rows: the 16-bytes-per-lane row kernel)
    anc = S.random_genome(40000, 31)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 700 + g) if g else anc, k)) for g in range(ngen)]
    allk = S.distinct(np.concatenate(gk))
    rng = np.random.default_rng(8)
    n = 70000
    batches = [np.ascontiguousarray(np.concatenate([allk[rng.integers(0, len(allk), n // 2)], S.snp_mutants(allk[rng.integers(0, len(allk), n // 2)], k, 3 + i)])) for i in range(3)]
    with BFT(k) as t:
        for g, km in enumerate(gk):
            t.insert_kmers(km, g)
        t.build()
        dev = torch.device("cuda", 0)
        rb = (ngen + 7) // 8
        expect = []
        for q in batches:  # the host calls (through row numbers / the three-launch path where they take it): what every replay must give
            b, off, ids = t.query_colors(q)
            b2, rows = t.query_color_rows(q)
            expect.append((b, off, ids, rows))
        cap = max(len(e[2]) for e in expect) + 16
        dq = torch.from_numpy(batches[0]).to(dev)
        bits = torch.zeros((n + 63) // 64 * 8, dtype=torch.uint8, device=dev)
        offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        ids = torch.zeros(cap, dtype=torch.int32, device=dev)
        need = torch.zeros(1, dtype=torch.int64, device=dev)
        rows = torch.zeros((n, rb), dtype=torch.uint8, device=dev)
        scr = torch.zeros(n, dtype=torch.int32, device=dev)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):  # direct calls first: scratch sized, bitmap dictionary derived
            t.query_colors_dev(dq.data_ptr(), n, bits.data_ptr(), offs.data_ptr(), ids.data_ptr(), cap, need.data_ptr(), s.cuda_stream)
            t.query_color_rows_dev(dq.data_ptr(), n, bits.data_ptr(), rows.data_ptr(), scr.data_ptr(), s.cuda_stream)
        s.synchronize()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=s):
            t.query_colors_dev(dq.data_ptr(), n, bits.data_ptr(), offs.data_ptr(), ids.data_ptr(), cap, need.data_ptr(), torch.cuda.current_stream().cuda_stream)
        with torch.cuda.graph(g2, stream=s):
            t.query_color_rows_dev(dq.data_ptr(), n, bits.data_ptr(), rows.data_ptr(), scr.data_ptr(), torch.cuda.current_stream().cuda_stream)
        for rep in range(6):
            q, (eb, eoff, eids, erows) = batches[rep % 3], expect[rep % 3]
            dq.copy_(torch.from_numpy(q))
            offs.zero_(); ids.zero_(); bits.zero_()
            g1.replay()
            torch.cuda.synchronize()
            assert (bits.cpu().numpy()[: len(eb)] == eb).all(), rep
            assert (offs.cpu().numpy().astype(np.uint64) == eoff).all(), rep
            assert int(need[0]) == len(eids) and (ids[: len(eids)].cpu().numpy().astype(np.uint32) == eids).all(), rep
            rows.zero_(); bits.zero_()
            g2.replay()
            torch.cuda.synchronize()
            assert (bits.cpu().numpy()[: len(eb)] == eb).all() and (rows.cpu().numpy() == erows).all(), rep
            if rep == 2:  # a direct call between the replays (its own scratch epoch, the handle's event)
                with torch.cuda.stream(s):
                    t.query_colors_dev(dq.data_ptr(), n, bits.data_ptr(), offs.data_ptr(), ids.data_ptr(), cap, need.data_ptr(), s.cuda_stream)
                s.synchronize()
                assert (offs.cpu().numpy().astype(np.uint64) == eoff).all()



def test_captured_sequence_queries_replay():
    """bft_gpu_query_sequences_dev recorded into a HIP graph (encode, plan, a multi-tile scan of the reads' position counts -- bracketed by zeroing
    kernels of the library's own under capture --, lookups, tally) and replayed on other reads in the same buffers: every replay gives the rows of a
    direct call.  (Enough reads for the scan to take its multi-tile form: more than 4096.)"""
    import torch
    from bloomfiltertrie_amd import BFT
    k, ngen, rl, n_reads = 27, 9, 80, 6000
    anc = S.random_genome(60000, 77)
    gs = [S.mutate(anc, 0.01, 900 + g) if g else anc for g in range(ngen)]
    with BFT(k) as t:
        for g, seq in enumerate(gs):
            t.insert_kmers(S.distinct(S.kmers_of(seq, k)), g)
        t.build()
        dev = torch.device("cuda", 0)
        rng = np.random.default_rng(4)
        def blob_of(seed):
            r = np.random.default_rng(seed)
            src = gs[seed % ngen]
            starts = r.integers(0, len(src) - rl, n_reads)
            codes = np.stack([src[a:a + rl] for a in starts])
            return np.ascontiguousarray(S._ASCII[codes]).reshape(-1)
        blobs = [blob_of(i) for i in range(3)]
        off = torch.arange(0, (n_reads + 1) * rl, rl, dtype=torch.int64, device=dev)
        d_blob = torch.from_numpy(blobs[0]).to(dev)
        rows = torch.zeros((n_reads, (ngen + 7) // 8), dtype=torch.uint8, device=dev)
        s = torch.cuda.Stream()
        expect = []
        for b in blobs:  # direct calls: the expected rows (and the handle's scratch sized for the capture)
            d_blob.copy_(torch.from_numpy(b))
            torch.cuda.synchronize()
            with torch.cuda.stream(s):
                t.query_sequences_dev(d_blob.data_ptr(), off.data_ptr(), n_reads, n_reads * rl, 0.8, rows.data_ptr(), False, s.cuda_stream)
            s.synchronize()
            expect.append(rows.cpu().numpy().copy())
        assert expect[0].any() and not (expect[0] == expect[1]).all()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            t.query_sequences_dev(d_blob.data_ptr(), off.data_ptr(), n_reads, n_reads * rl, 0.8, rows.data_ptr(), False, torch.cuda.current_stream().cuda_stream)
        for rep in range(6):
            d_blob.copy_(torch.from_numpy(blobs[rep % 3]))
            rows.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert (rows.cpu().numpy() == expect[rep % 3]).all(), rep
