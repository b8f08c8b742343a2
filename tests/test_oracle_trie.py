"""Oracle (CPU restatement) against mathematical ground truth: the BFT is an exact index, so
presence == set membership and colours == set of genomes that inserted the k-mer.
Edge cases follow SURVEY.md 8c (3): UC-only node, one CC, many CCs, CC >= 3584 prefixes (p=14/s=4),
suffix group of exactly 255 -> burst to child Node at 256, level_min==0 levels (k >= 18),
annotation modes 0/1/2, >= 64 genomes (multi-byte ids)."""
import numpy as np
import pytest

from bloomfiltertrie_amd import synth as S


def _check_presence(t, km, k, seed=0, extra=None):
    rng = np.random.default_rng(seed)
    qs = [km, S.pack_codes(rng.integers(0, 4, (max(1000, len(km) // 2), k), dtype=np.uint8)), S.snp_mutants(km, k, seed + 1)]
    if extra is not None:
        qs.append(extra)
    q = np.concatenate(qs)
    q = q[rng.permutation(len(q))]
    bits = t.query_presence(q)
    truth = S.member(q, km)
    assert (S.from_bits(bits, len(q)) == truth).all()
    bits_mt = t.query_presence(q, threads=3)
    assert (bits_mt == bits).all()
    return q, truth


@pytest.mark.parametrize("k", [9, 18, 27, 36, 45, 63])
def test_presence_random_genome(oracle_mod, k):
    g = S.random_genome(60000, 10 + k)
    km = S.distinct(S.kmers_of(g, k))
    t = oracle_mod.OracleBFT(k)
    t.insert_kmers(km, 0)
    assert t.stats()["kmers"] == len(km)
    _check_presence(t, km, k)
    ek, _ = t.extract()
    assert sorted(S.row_keys(ek).tolist()) == sorted(S.row_keys(km).tolist())


@pytest.mark.parametrize("n", [0, 1, 254, 255, 256, 300])
def test_uc_and_first_burst(oracle_mod, n):
    k = 27
    km = S.distinct(S.pack_codes(np.random.default_rng(n).integers(0, 4, (n, k), dtype=np.uint8))) if n else np.zeros((0, 7), np.uint8)
    t = oracle_mod.OracleBFT(k)
    t.insert_kmers(km, 0)
    st = t.stats()
    if len(km) < 255:
        assert st["ccs"] == 0 and st["root_uc_rows"] == len(km)  # src/insertNode.c:183-192
    else:
        assert st["ccs"] == 1  # src/insertNode.c:197-223
    _check_presence(t, km, k, extra=np.zeros((3, 7), np.uint8))


@pytest.mark.parametrize("k,levels", [(18, 1), (27, 1), (27, 2), (36, 2), (36, 3), (63, 3), (45, 4)])
def test_deep_tries(oracle_mod, k, levels):
    km = S.low_entropy_kmers(60000, k, 24, seed=k * 7 + levels, levels=levels)
    t = oracle_mod.OracleBFT(k)
    t.insert_kmers(km, 0)
    st = t.stats()
    assert st["kmers"] == len(km)
    assert st["child_nodes"] > 0  # suffix groups > 255 burst into child Nodes (src/insertNode.c:291)
    _check_presence(t, km, k)


@pytest.mark.parametrize("k,levels", [(27, 2), (36, 3), (63, 3)])
def test_level_min_0_cluster_walk(oracle_mod, k, levels):
    """Levels whose suffix length is not 9 mod 36 have no extra_filter3 in the reference: findCluster walks children_type from the
    SkipFilter3 cell and hands its running child / node counts to presenceKmer (src/presenceNode.c:1690-1812, :1425-1448), restated
    as findCluster_lm0.  A trie with many prefixes per cluster and child Nodes on such levels answers exactly (ground truth), its
    colours come from the rows those counts locate, and the counting mode sees the walk (more bytes per query than levels alone)."""
    rng = np.random.default_rng(k)
    km = S.low_entropy_kmers(150000, k, 6, seed=3 + k, levels=levels)   # few root prefixes: big nodes below the root
    halves = np.array_split(km[rng.permutation(len(km))], 2)
    o = oracle_mod.OracleBFT(k)
    c = oracle_mod.OracleBFT(k, count=True)
    for g, part in enumerate(halves):
        o.insert_kmers(np.ascontiguousarray(part), g)
        c.insert_kmers(np.ascontiguousarray(part), g)
    st = o.stats()
    assert st["child_nodes"] > 0 and st["ccs"] > st["root_ccs"]
    q = np.concatenate([km, S.snp_mutants(km, k, 2), S.snp_mutants(km[::2], k, 5)])
    bits, off, ids = o.query_colors(q)
    truth = S.member(q, km)
    assert (S.from_bits(bits, len(q)) == truth).all()
    which = {row.tobytes(): g for g, part in enumerate(halves) for row in part}
    sizes = np.diff(off)
    for i in np.flatnonzero(truth)[:: max(1, int(truth.sum()) // 4000)]:
        assert sizes[i] == 1 and ids[off[i]] == which[q[i].tobytes()]
    cbits, cnt = c.query_presence_count(q)
    assert (cbits == bits).all() and cnt["levels"] > len(q) and cnt["bytes"] > 40 * len(q)


def test_group_exactly_255_then_256(oracle_mod):
    k = 18
    rng = np.random.default_rng(3)
    base = rng.integers(0, 4, (1, 9), dtype=np.uint8)
    suf = S.distinct(S.pack_codes(rng.integers(0, 4, (4000, 9), dtype=np.uint8)))
    sufc = S.unpack_codes(suf, 9)[:300]
    filler = S.distinct(S.pack_codes(rng.integers(0, 4, (400, k), dtype=np.uint8)))
    same = S.pack_codes(np.concatenate([np.repeat(base, 300, 0), sufc], axis=1))
    t = oracle_mod.OracleBFT(k)
    t.insert_kmers(filler, 0)          # creates the first CC
    t.insert_kmers(same[:1], 0)        # prefix may land in CC or UC
    t.insert_kmers(same[1:255], 0)
    km = np.concatenate([filler, same[:255]])
    _check_presence(t, km, k)
    n_before = t.stats()["child_nodes"]
    t.insert_kmers(same[255:300], 0)
    km = np.concatenate([filler, same[:300]])
    _check_presence(t, km, k)
    assert t.stats()["kmers"] == len(S.distinct(km))
    assert t.stats()["child_nodes"] >= n_before


def test_cc_reaches_p14_mode(oracle_mod):
    k = 27
    g = S.random_genome(400000, 99)
    km = S.distinct(S.kmers_of(g, k))
    t = oracle_mod.OracleBFT(k)
    t.insert_kmers(km, 0)
    st = t.stats()
    assert st["ccs_s4"] > 0 and st["ccs_s4"] < st["ccs"]  # both filter2 geometries (src/insertNode.c:134-135)
    _check_presence(t, km, k)


def _colour_truth(genome_kmers):
    d = {}
    for gid, km in enumerate(genome_kmers):
        for key in S.row_keys(km).tolist():
            d.setdefault(key, []).append(gid)
    return d


@pytest.mark.parametrize("k,ngen", [(27, 6), (18, 10), (36, 70), (27, 200)])
def test_colours(oracle_mod, k, ngen):
    anc = S.random_genome(3000 if ngen > 20 else 20000, 5)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 100 + g) if g else anc, k)) for g in range(ngen)]
    # make some genomes skip ranges to exercise range mode (mode 1) and list mode (mode 2)
    t = oracle_mod.OracleBFT(k)
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
        t.insert_kmers(km[:10], g)  # re-inserting the same genome id is a no-op (last_added == id)
    truth = _colour_truth(gk)
    allk = S.distinct(np.concatenate(gk))
    rng = np.random.default_rng(1)
    q = np.concatenate([allk[rng.permutation(len(allk))[:5000]], S.pack_codes(rng.integers(0, 4, (500, k), dtype=np.uint8))])
    bits, off, ids = t.query_colors(q)
    keys = S.row_keys(q).tolist()
    pres = S.from_bits(bits, len(q))
    for i, key in enumerate(keys):
        exp = truth.get(key, [])
        assert pres[i] == bool(exp)
        assert ids[int(off[i]):int(off[i + 1])].tolist() == exp
    # extraction: every k-mer with its colour set
    ek, ecs = t.extract()
    assert len(ek) == len(allk)
    for key, cs in list(zip(S.row_keys(ek).tolist(), ecs.tolist()))[:2000]:
        assert t.colorset(cs) == truth[key]


def test_annotation_codec_roundtrip(oracle_mod):
    O = oracle_mod
    rng = np.random.default_rng(0)
    cases = [[0], [1], [5], [63], [64], [0, 1, 2, 3], list(range(0, 100)), [3, 64, 4095, 4096, 70000],
             list(range(10, 20)) + list(range(100, 3000)), [0, 2, 4, 6, 8], [99999999]]
    for _ in range(200):
        n = int(rng.integers(1, 40))
        cases.append(sorted(set(rng.integers(0, int(rng.choice([8, 70, 5000, 300000])), n).tolist())))
    for ids in cases:
        b = O.annot_encode(ids)
        assert O.annot_decode(b) == ids, ids
        assert O.annot_decode(b + b"\x00\x00") == ids  # zero padding up to size_annot is harmless
        # size = the minimum of the three modes (src/annotation.c:620-650)
        s0 = (3 + ids[-1] + 7) // 8
        s2 = sum(O.nb_bytes_id(i) for i in ids)
        runs, a = [], 0
        while a < len(ids):
            b2 = a
            while b2 + 1 < len(ids) and ids[b2 + 1] == ids[b2] + 1:
                b2 += 1
            runs.append((ids[a], ids[b2]))
            a = b2 + 1
        s1 = sum(O.nb_bytes_id(x) + O.nb_bytes_id(y) for x, y in runs)
        assert len(b) == min(s0, s1, s2)


def test_rejects_k_not_multiple_of_9(oracle_mod):
    for k in (31, 8, 135, 0):
        with pytest.raises(ValueError):
            oracle_mod.OracleBFT(k)


@pytest.mark.parametrize("k", [9, 18, 27, 36, 63])
def test_branching_against_ground_truth(oracle_mod, k):
    """-query_branching semantics (src/file_io.c:943-998): >1 successor or >1 predecessor among stored k-mers."""
    anc = S.random_genome(4000, 3)
    genomes = [anc, S.mutate(anc, 0.05, 1), S.mutate(anc, 0.05, 2)]
    km = S.distinct(np.concatenate([S.kmers_of(g, k) for g in genomes]))
    t = oracle_mod.OracleBFT(k)
    t.insert_kmers(km, 0)
    q = np.concatenate([km, S.snp_mutants(km[::3], k, 5)])
    bits, counts, nbr = t.query_branching(q)
    present = set(map(bytes, km))
    codes = S.unpack_codes(q, k)
    for i in range(0, len(q), 7):
        c = codes[i]
        succ = sum(bytes(S.pack_codes(np.concatenate([c[1:], [x]])[None, :])[0]) in present for x in range(4))
        pred = sum(bytes(S.pack_codes(np.concatenate([[x], c[:-1]])[None, :])[0]) in present for x in range(4))
        assert counts[i] == (succ << 4 | pred), (i, counts[i], succ, pred)
        assert bool(bits[i >> 3] >> (i & 7) & 1) == (succ > 1 or pred > 1)
    assert nbr == int(np.unpackbits(bits, bitorder="little")[: len(q)].sum())


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


@pytest.mark.parametrize("canonical", [False, True])
def test_query_sequence_against_ground_truth(oracle_mod, canonical):
    """query_sequence (src/bft.c:1241-1351): genomes holding >= ceil(nb_kmers * threshold) of a read's k-mers."""
    import math
    k, ngen = 27, 5
    anc = S.random_genome(6000, 12)
    genomes = [S.mutate(anc, 0.03, 40 + g) for g in range(ngen)]
    strs = ["".join("ACGT"[c] for c in g) for g in genomes]
    t = oracle_mod.OracleBFT(k)
    sets = []
    for g, s in enumerate(strs):
        kms = {s[i:i + k] for i in range(len(s) - k + 1)}
        if canonical:
            kms = {min(x, _revcomp(x)) for x in kms}
        sets.append(kms)
        packed, valid = S.ascii_to_packed(sorted(kms), k)
        t.insert_kmers(packed, g)
    rng = np.random.default_rng(3)
    reads = []
    for _ in range(40):
        g = int(rng.integers(0, ngen))
        a = int(rng.integers(0, len(strs[g]) - 200))
        r = strs[g][a:a + int(rng.integers(k, 200))]
        if rng.random() < 0.5:
            r = _revcomp(r)
        if rng.random() < 0.3:
            r = r[:10] + "N" + r[11:]
        reads.append(r)
    reads += ["ACGT", "A" * k]
    for thr in (0.25, 0.8, 1.0):
        for r in reads:
            nb = max(0, len(r) - k + 1)
            mn = math.ceil(nb * thr)
            exp = []
            for g in range(ngen):
                c = 0
                for i in range(nb):
                    x = r[i:i + k]
                    if set(x) - set("ACGT"):
                        continue
                    if canonical and x >= _revcomp(x):
                        x = _revcomp(x)
                    c += x in sets[g]
                if c and c >= mn:
                    exp.append(g)
            assert t.query_sequence(r, thr, canonical, ngen) == exp, (r, thr)
