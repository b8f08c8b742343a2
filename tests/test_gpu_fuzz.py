"""Randomised cross-check on the GPU box: many seeded configurations (k, sizes, entropy, genomes, insertion batches),
HIP path vs oracle for presence, colours, branching and extraction."""
import numpy as np
import pytest

from bloomfiltertrie_amd import synth as S

import os

pytestmark = pytest.mark.gpu
N_SEEDS = int(os.environ.get("BFT_FUZZ_SEEDS", "24"))  # BFT_FUZZ_SEEDS=400 for a soak run


@pytest.mark.parametrize("seed", list(range(N_SEEDS)))
def test_fuzz(oracle_mod, seed):
    from bloomfiltertrie_amd import BFT
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([9, 18, 27, 36, 45, 54, 63, 72, 81, 90, 99, 108, 117, 126]))
    ngen = int(rng.integers(1, 9))
    levels = int(rng.integers(0, min(4, k // 9) + 1))
    n = int(rng.choice([0, 1, 200, 254, 255, 256, 257, 3000, 40000]))
    if levels and n:
        base = S.low_entropy_kmers(n, k, int(rng.integers(2, 40)), seed=seed, levels=levels)
    elif n:
        base = S.distinct(S.pack_codes(rng.integers(0, 4, (n, k), dtype=np.uint8)))
    else:
        base = np.zeros((0, S.kmer_bytes(k)), np.uint8)
    t, o = BFT(k), oracle_mod.OracleBFT(k)
    t.set_option("flat_min", int(rng.choice([3584, 3584, 65536, 300, 1])))  # flat form of the CCs: default rule, off, small CCs too
    for g in range(ngen):
        km = base[rng.random(len(base)) < rng.uniform(0.2, 1.0)] if len(base) else base
        for part in np.array_split(km, int(rng.integers(1, 4))):
            t.insert_kmers(np.ascontiguousarray(part), g)
        o.insert_kmers(np.ascontiguousarray(km), g)
        if rng.random() < 0.4:
            t.build()  # incremental rebuilds at random points
    parts = [S.pack_codes(rng.integers(0, 4, (int(rng.integers(1, 500)), k), dtype=np.uint8))]
    if len(base):
        parts += [base[:: max(1, len(base) // 3000)], S.snp_mutants(base[:: max(1, len(base) // 2000)], k, seed)]
    q = np.concatenate(parts)
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    t.set_option("query_probe", int(rng.choice([0, 4, 8])))  # suffix-group probe mode: a tuning knob, same answers
    bits, off, ids = t.query_colors(q)
    obits, ooff, oids = o.query_colors(q)
    assert (bits == obits).all() and (off == ooff).all() and (ids == oids).all()
    assert (t.query_presence(q) == obits).all()
    if ngen:  # fixed-width rows and (row, colour set) locations agree with the id lists
        _, rows = t.query_color_rows(q)
        unp = np.unpackbits(rows, axis=1, bitorder="little")[:, :ngen]
        exp = np.zeros_like(unp)
        for i in range(len(q)):
            exp[i, oids[int(ooff[i]):int(ooff[i + 1])]] = 1
        assert (unp == exp).all()
        b3, rws, sets = t.query_rows(q)
        assert (b3 == obits).all()
        pres = S.from_bits(obits, len(q)).astype(bool)
        assert (rws[~pres] == 0xFFFFFFFF).all() and (sets[pres] != 0xFFFFFFFF).all()
    bb, bc = t.query_branching(q, with_counts=True)
    ob, oc, _ = o.query_branching(q)
    assert (bc == oc).all() and (bb == ob).all()
    t.set_option("flat_min", int(rng.choice([1, 100, 3584, 65536])))  # re-derive the flat arrays of the built image
    assert (t.query_presence(q) == obits).all()
    ek, _ = t.extract()
    ok, _ = o.extract()
    assert sorted(map(bytes, ek)) == sorted(map(bytes, ok))


def _keys64(packed):
    pad = np.zeros((len(packed), 8), dtype=np.uint8)
    pad[:, :packed.shape[1]] = packed
    return pad.view(np.uint64).reshape(-1)


@pytest.mark.parametrize("seed", list(range(6)))
def test_fuzz_large_groups_against_ground_truth(seed):
    """Bigger indexes than the oracle-checked fuzz: suffix groups of up to 255 rows, child Nodes, remainder groups (k = 31),
    every probe mode and residency; presence and branching counts against set arithmetic on the integer form of the k-mers."""
    from bloomfiltertrie_amd import BFT
    rng = np.random.default_rng(77 + seed)
    k = int(rng.choice([27, 31, 18]))
    n = int(rng.choice([300_000, 1_000_000]))
    if seed % 2:
        base = S.low_entropy_kmers(n, k, int(rng.integers(3, 400)), seed=seed, levels=int(rng.integers(1, k // 9 + 1)))
    else:
        base = S.distinct(S.kmers_of(S.random_genome(n, 100 + seed), k))
    t = BFT(k)
    t.insert_kmers(base[: len(base) // 2], 0)
    t.insert_kmers(base[len(base) // 3:], 1)
    keys = np.unique(_keys64(base))
    q = np.concatenate([base[:: max(1, len(base) // 200_000)], S.snp_mutants(base[:: max(1, len(base) // 200_000)], k, seed),
                        S.pack_codes(rng.integers(0, 4, (5000, k), dtype=np.uint8))])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    qk = _keys64(q)
    truth = np.isin(qk, keys)
    mask = np.uint64((1 << (2 * k)) - 1)
    succ = np.zeros(len(q), dtype=np.int64)
    pred = np.zeros(len(q), dtype=np.int64)
    for nt in range(4):
        succ += np.isin((qk >> np.uint64(2)) | np.uint64(nt << (2 * (k - 1))), keys)
        pred += np.isin(((qk << np.uint64(2)) & mask) | np.uint64(nt), keys)
    exp_counts = ((succ << 4) | pred).astype(np.uint8)
    for wgs, probe in [(0, 0), (1, 4), (2, 8), (1, 8), (2, 4)]:
        t.set_option("query_wgs_per_cu", wgs)
        t.set_option("query_probe", probe)
        assert (S.from_bits(t.query_presence(q), len(q)).astype(bool) == truth).all(), (k, n, wgs, probe)
        bb, bc = t.query_branching(q, with_counts=True)
        assert (bc == exp_counts).all(), (k, n, wgs, probe)
        assert (S.from_bits(bb, len(q)).astype(bool) == ((succ > 1) | (pred > 1))).all()
        assert (t.query_branching(q) == bb).all()  # early-exit variant
