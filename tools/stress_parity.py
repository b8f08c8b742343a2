#!/usr/bin/env python3
"""Randomised end-to-end check against ground truth kept in Python dictionaries: random k, genome counts, insertion orders (ascending /
shuffled ids, duplicate batches, incremental rebuilds), then presence, colour sets, colour rows (host and device calls), branching, sequence
queries (host and device calls) -- across the build paths (composite / narrow-id / general sort) and the derived tables.
usage: stress_parity.py [rounds] [seed] [first round]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, synth as S, _lib as L  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
verbose = bool(os.environ.get("STRESS_VERBOSE"))
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for r in range(first, rounds):
    rng = np.random.default_rng(seed0 * 1000 + r)
    k = int(rng.choice([13, 18, 21, 27, 27, 31, 31, 32, 36, 45, 63, 64, 72, 99, 126]))
    ngen = int(rng.choice([1, 2, 4, 5, 9, 40, 130, 300]))
    glen = int(rng.integers(3000, 30000))
    deep = min(int(rng.integers(0, 3)), max(0, k // 9 - 1))
    if deep:
        base = S.low_entropy_kmers(glen, k, 12, seed=r + 7, levels=deep)
        s = None
    else:
        g = S.random_genome(glen, 100 + r)
        s = "".join("ACGT"[c] for c in g)
        base = S.distinct(S.kmers_of(g, k))
    nb = len(base)
    member = rng.random((ngen, nb)) < rng.uniform(0.05, 0.9)
    member[int(rng.integers(0, ngen)), :] |= rng.random(nb) < 0.5
    order = np.arange(ngen)
    shuffled = bool(rng.random() < 0.3)
    if shuffled:
        rng.shuffle(order)
    t = BFT(k)
    if verbose:
        print(f"round {r} starts: k={k} genomes={ngen} glen={glen} deep={deep} shuffled={shuffled}", flush=True)
        _so = t.set_option
        t.set_option = lambda n_, v_: (print("  option", n_, v_, flush=True), _so(n_, v_))[1]
    if rng.random() < 0.3:
        t.set_option("kmer_hash", 0)
    if rng.random() < 0.3:
        t.set_option("build_composite", 0)
    if rng.random() < 0.5:
        t.set_option("build_msd", int(rng.choice([0, 2])))          # one device-wide sort / root-prefix buckets at any size
    if rng.random() < 0.5:
        t.set_option("flush_pairs", int(rng.choice([1024, 5000, 40000])))  # the log is merged into the index many times on the way
    if rng.random() < 0.3:
        t.set_option("kmer_hash_load", int(rng.choice([20, 65, 80])))
    if rng.random() < 0.4:
        t.set_option("compact_table", 1)                                   # the sorted table leaves HBM whenever the k-mer hash can stand in for it
    for ov in os.environ.get("STRESS_OPTS", "").split():  # (bisecting a failure: options applied on top of the round's own)
        t.set_option(ov.split("=")[0], int(ov.split("=")[1]))
    cut = int(rng.integers(0, ngen + 1))
    use_async = bool(rng.random() < 0.4)
    for j, gi in enumerate(order):
        if j == cut:
            t.build()  # incremental: the sorted store is merged with the log of the rest
        rows_g = np.ascontiguousarray(base[member[gi]])
        if len(rows_g):
            if use_async:  # stream-ordered device insert: the tensor goes back to the allocator at once
                d_rows_g = torch.from_numpy(rows_g).to(dev)
                t.insert_kmers_dev_async(d_rows_g.data_ptr(), len(rows_g), int(gi), st)
                del d_rows_g
            else:
                t.insert_kmers(rows_g, int(gi))
            if rng.random() < 0.2:
                t.insert_kmers(np.ascontiguousarray(rows_g[::3]), int(gi))  # duplicates
    present_any = member.any(axis=0)
    q = np.concatenate([base, S.snp_mutants(base[::3], k, r + 1)])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    pos = {key: i for i, key in enumerate(S.row_keys(base).tolist())}
    idx = np.array([pos.get(key, -1) for key in S.row_keys(q).tolist()])
    exp_pres = (idx >= 0) & present_any[np.maximum(idx, 0)]
    bits, off, ids = t.query_colors(q)
    pres = S.from_bits(bits, len(q)).astype(bool)
    assert (pres == exp_pres).all(), ("presence", r, k, ngen)
    for i in rng.integers(0, len(q), 300):
        exp = np.flatnonzero(member[:, idx[i]]).tolist() if exp_pres[i] else []
        assert ids[int(off[i]):int(off[i + 1])].tolist() == exp, ("colours", r, k, ngen, i)
    exp_rows = np.zeros((len(q), ngen), dtype=np.uint8)
    exp_rows[exp_pres] = member[:, idx[exp_pres]].T
    b2, rows = t.query_color_rows(q)
    assert (np.unpackbits(rows, axis=1, bitorder="little")[:, :ngen] == exp_rows).all(), ("rows host", r, k, ngen)
    dq = torch.from_numpy(q).to(dev)
    dbits = torch.zeros(((len(q) + 63) // 64) * 8, dtype=torch.uint8, device=dev)
    drows = torch.zeros((len(q), (ngen + 7) // 8), dtype=torch.uint8, device=dev)
    dscr = torch.zeros(len(q), dtype=torch.int32, device=dev)
    L.check(t._lib.bft_gpu_query_color_rows_dev(t._h, dq.data_ptr(), len(q), dbits.data_ptr(), drows.data_ptr(), dscr.data_ptr(), st))
    torch.cuda.synchronize()
    assert (np.unpackbits(drows.cpu().numpy(), axis=1, bitorder="little")[:, :ngen] == exp_rows).all(), ("rows dev", r, k, ngen)
    assert (S.from_bits(dbits.cpu().numpy(), len(q)).astype(bool) == exp_pres).all()
    assert (S.from_bits(t.query_presence(q), len(q)).astype(bool) == exp_pres).all()  # (after the colour-set launch: rows mode is back)
    # branching (src/file_io.c:943-998): successors / predecessors counted among the stored k-mers, whether or not the k-mer itself is stored
    stored_keys = set(S.row_keys(np.ascontiguousarray(base[present_any])).tolist())
    qs = np.ascontiguousarray(q[rng.integers(0, len(q), 400)])
    codes = S.unpack_codes(qs, k)
    nbr = np.zeros((len(qs), 2), dtype=np.int64)
    for c in range(4):
        suc = np.concatenate([codes[:, 1:], np.full((len(qs), 1), c, np.uint8)], axis=1)
        pre = np.concatenate([np.full((len(qs), 1), c, np.uint8), codes[:, :-1]], axis=1)
        nbr[:, 0] += np.array([key in stored_keys for key in S.row_keys(S.pack_codes(suc)).tolist()])
        nbr[:, 1] += np.array([key in stored_keys for key in S.row_keys(S.pack_codes(pre)).tolist()])
    bb, cnt = t.query_branching(qs, with_counts=True)
    assert (cnt == ((nbr[:, 0] << 4) | nbr[:, 1])).all(), ("branching counts", r, k, ngen)
    assert (S.from_bits(bb, len(qs)).astype(bool) == ((nbr[:, 0] > 1) | (nbr[:, 1] > 1))).all(), ("branching bits", r, k, ngen)
    assert (S.from_bits(t.query_branching(qs), len(qs)).astype(bool) == ((nbr[:, 0] > 1) | (nbr[:, 1] > 1))).all(), ("branching bits, no counts", r, k, ngen)
    b3, rws, sets = t.query_rows(q)
    ek, ecs = t.extract()
    assert (ek[rws[exp_pres]] == q[exp_pres]).all() and (ecs[rws[exp_pres]] == sets[exp_pres]).all(), ("rows", r, k, ngen)
    if s is not None:
        reads, spans = [], []
        for _ in range(40):
            a = int(rng.integers(0, glen - 200))
            n = int(rng.integers(max(1, k - 2), 200))
            reads.append(s[a:a + n])
            spans.append((a, max(0, n - k + 1)))
        thr = float(rng.choice([0.2, 0.75, 1.0]))
        got = t.query_sequences(reads, thr)
        kpos = np.array([pos[key] for key in S.row_keys(S.kmers_of(g, k)).tolist()])  # position in the genome -> row of base
        for (a, m), gl in zip(spans, got):
            cnt = member[:, kpos[a:a + m]].sum(axis=1) if m else np.zeros(ngen, dtype=int)
            assert gl == np.flatnonzero((cnt >= math.ceil(m * thr)) & (cnt > 0)).tolist(), ("sequences", r, k, ngen)
        enc = [x.encode() for x in reads]
        offs = np.zeros(len(enc) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(e) for e in enc])
        d_blob = torch.from_numpy(np.frombuffer(b"".join(enc), dtype=np.uint8).copy()).to(dev)
        d_off = torch.from_numpy(offs).to(dev)
        d_r = torch.zeros((len(enc), (ngen + 7) // 8), dtype=torch.uint8, device=dev)
        t.query_sequences_dev(d_blob.data_ptr(), d_off.data_ptr(), len(enc), int(offs[-1]), thr, d_r.data_ptr(), False, st)
        torch.cuda.synchronize()
        unp = np.unpackbits(d_r.cpu().numpy(), axis=1, bitorder="little")[:, :ngen]
        assert [np.flatnonzero(x).tolist() for x in unp] == got, ("sequences dev", r, k, ngen)
    info = t.info()
    assert info["kmers"] == int(present_any.sum())
    print(f"round {r}: k={k} genomes={ngen} kmers={info['kmers']} deep={deep} shuffled={shuffled} cut={cut} async={use_async} ok", flush=True)
    t.close()
print("stress OK")
