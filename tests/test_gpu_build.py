"""GPU bulk build (k_pack_to_tform, radix sort, k_flags/k_scatter, bft_assemble.hip kernels) against the host
restatement of the same construction (csrc/bft_index.cpp, test helper library): every array of the image must be
bit-identical, on shallow, deep, tiny and multi-word-k inputs."""
import ctypes as C
import os

import numpy as np
import pytest

from bloomfiltertrie_amd import BFT, _lib, synth as S

pytestmark = pytest.mark.gpu

ARRAYS = ["tk", "nodes", "bfT", "ccs", "f2w", "clus", "child", "uck", "ucrow", "ccx", "f18", "fent"]


@pytest.fixture(scope="module")
def hostlib():
    lib = C.CDLL(os.path.join(_lib.CSRC, "libbft_hosttest.so"))
    lib.bft_hosttest_build.restype = C.c_void_p
    lib.bft_hosttest_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int]
    lib.bft_hosttest_get_array.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.bft_hosttest_free.argtypes = [C.c_void_p]
    return lib


def _host_arrays(hostlib, km, k):
    km = np.ascontiguousarray(km)
    h = hostlib.bft_hosttest_build(km.ctypes.data, len(km), k, 0, 0)
    assert h
    out = {}
    for name in ARRAYS:
        n = C.c_uint64()
        assert hostlib.bft_hosttest_get_array(h, name.encode(), None, 0, C.byref(n)) == 0
        buf = np.zeros(n.value, dtype=np.uint8)
        assert hostlib.bft_hosttest_get_array(h, name.encode(), buf.ctypes.data, n.value, C.byref(n)) == 0
        out[name] = buf
    hostlib.bft_hosttest_free(h)
    return out


def _compare(hostlib, km, k, pieces=1):
    from bloomfiltertrie_amd import BFT
    t = BFT(k)
    for part in np.array_split(km, pieces):  # several insert batches, with duplicates across them
        t.insert_kmers(part, 0)
    t.insert_kmers(km[: len(km) // 3], 0)
    t.build()
    host = _host_arrays(hostlib, km, k)
    for name in ARRAYS:
        got = t.debug_array(name)
        assert got.shape == host[name].shape, (name, got.shape, host[name].shape)
        assert (got == host[name]).all(), name
    return t.info()


@pytest.mark.parametrize("k", [9, 18, 27, 36, 63, 126, 31, 13, 40, 17, 100])
def test_random_genome(hostlib, k):
    info = _compare(hostlib, S.distinct(S.kmers_of(S.random_genome(300000, 3 + k), k)), k, pieces=3)
    assert info["ccs"] > 1


@pytest.mark.parametrize("k,levels", [(18, 1), (27, 1), (27, 2), (36, 3), (63, 3), (45, 4), (31, 2), (31, 3), (22, 1), (40, 4)])
def test_deep(hostlib, k, levels):
    info = _compare(hostlib, S.low_entropy_kmers(200000, k, 24, seed=k + levels, levels=levels), k, pieces=2)
    assert info["child_nodes"] > 0


@pytest.mark.parametrize("n", [0, 1, 254, 255, 256, 300, 5000])
def test_tiny(hostlib, n):
    km = S.distinct(S.pack_codes(np.random.default_rng(n).integers(0, 4, (n, 27), dtype=np.uint8))) if n else np.zeros((0, 7), np.uint8)
    _compare(hostlib, km, 27)


def test_large_root(hostlib):
    info = _compare(hostlib, S.distinct(S.kmers_of(S.random_genome(3000000, 5), 27)), 27, pieces=4)
    assert 0 < info["ccs_s4"] <= info["ccs"]


def test_many_small_nodes(hostlib):
    # thousands of child nodes at depth 1 and 2
    km = S.low_entropy_kmers(1500000, 27, 3000, seed=11, levels=1)
    info = _compare(hostlib, km, 27)
    assert info["child_nodes"] > 1000


def _rk(n, k, seed):
    return S.distinct(S.kmers_of(S.random_genome(n + k - 1, seed), k))


def test_image_pack_unpack_roundtrip(tmp_path):
    """bft_gpu_image_pack -> bft_gpu_image_unpack gives an index with the same answers, the same .bft bytes, and
    insertion continues on the copy exactly as on the original."""
    import torch
    k = 27
    g0 = _rk(30000, k, 5)
    g1 = np.concatenate([g0[:10000], _rk(15000, k, 6)])
    a = BFT(k)
    a.add_genome("first.fa")
    a.insert_kmers(g0, 0)
    a.add_genome("second.fa")
    a.insert_kmers(g1, 1)
    a.build()
    n = a.image_size()
    blob = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    a.image_pack(blob.data_ptr(), n)
    b = BFT.from_image(blob.data_ptr(), n, device=0)
    del blob
    ia, ib = a.info(), b.info()
    assert ia == ib
    q = np.concatenate([g0[::3], g1[::3], _rk(20000, k, 7)])
    assert (a.query_presence(q) == b.query_presence(q)).all()
    ba, ra = a.query_color_rows(q)
    bb, rb = b.query_color_rows(q)
    assert (ba == bb).all() and (ra == rb).all()
    assert (a.query_branching(q[:5000]) == b.query_branching(q[:5000])).all()
    a.write_bft(str(tmp_path / "a.bft"))
    b.write_bft(str(tmp_path / "b.bft"))
    assert open(tmp_path / "a.bft", "rb").read() == open(tmp_path / "b.bft", "rb").read()
    g2 = _rk(5000, k, 8)
    for t in (a, b):
        t.add_genome("third.fa")
        t.insert_kmers(np.concatenate([g2, g0[:100]]), 2)
        t.build()
    assert a.info() == b.info()
    q2 = np.concatenate([g2, g0[:200]])
    assert (a.query_color_rows(q2)[1] == b.query_color_rows(q2)[1]).all()
    ka, ca = a.extract()
    kb, cb = b.extract()
    assert (ka == kb).all() and (ca == cb).all()


def test_image_unpack_rejects_garbage():
    import torch
    blob = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(Exception):
        BFT.from_image(blob.data_ptr(), 4096, device=0)
    with pytest.raises(Exception):
        BFT.from_image(blob.data_ptr(), 16, device=0)


def test_replicate_image_single_rank_rccl():
    """dist.replicate_image over the nccl (= RCCL) backend with world_size 1: the broadcast path end to end."""
    import torch
    import torch.distributed as dist
    from bloomfiltertrie_amd.dist import replicate_image
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        a = BFT(18)
        km = _rk(5000, 18, 3)
        a.insert_kmers(km, 0)
        a.build()
        assert replicate_image(a, 0, src=0) is a
        b = replicate_image(a, 0, src=0, always_copy=True)
        q = np.concatenate([km[::2], _rk(3000, 18, 4)])
        assert b is not a and (a.query_presence(q) == b.query_presence(q)).all() and a.info() == b.info()
    finally:
        dist.destroy_process_group()


def test_sharded_queries_single_rank_rccl():
    """dist.query_presence_sharded / query_color_rows_sharded / query_branching_sharded over the nccl (= RCCL) backend with
    world_size 1: the device-resident branch of each (pinned copy in, *_dev call, gathers out of HBM) equals the host entry points."""
    import torch
    import torch.distributed as dist
    from bloomfiltertrie_amd.dist import query_branching_sharded, query_color_rows_sharded, query_presence_sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29612")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        k = 27
        anc = S.random_genome(30000, 2)
        t = BFT(k)
        for g in range(19):
            t.insert_kmers(S.distinct(S.kmers_of(S.mutate(anc, 0.02, 70 + g), k)), g)
        t.build()
        ek, _ = t.extract()
        q = np.ascontiguousarray(np.concatenate([ek[::3], S.snp_mutants(ek[::5], k, 3)])[:20001])
        assert (query_presence_sharded(t, q) == t.query_presence(q)).all()
        bits, rows = query_color_rows_sharded(t, q, t.info()["genomes"])
        hb, hr = t.query_color_rows(q)
        assert (bits == hb).all() and rows.shape == hr.shape and (rows == hr).all()
        bb, counts = query_branching_sharded(t, q)
        xb, xc = t.query_branching(q, with_counts=True)
        assert (bb == xb).all() and (counts == xc).all()
        t.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("flat_min", [1, 255, 3584, 65536])
def test_flat_form_matches_host_restatement(hostlib, flat_min):
    """The derived arrays (extended CC headers, flat prefix bitmaps + ranks, flat entries) for any threshold are
    bit-identical to bft_flatten_index on the host, and the answers do not depend on the threshold."""
    k = 27
    km = np.concatenate([S.low_entropy_kmers(60000, k, 6, seed=3, levels=2), _rk(200000, k, 9)])
    km = S.distinct(km)
    t = BFT(k)
    t.insert_kmers(km, 0)
    t.build()
    q = np.concatenate([km[::7], S.snp_mutants(km[::11], k, 5), _rk(5000, k, 10)])
    ref_bits = t.query_presence(q)
    t.set_option("flat_min", flat_min)
    assert (t.query_presence(q) == ref_bits).all()
    hostlib.bft_hosttest_flatten.argtypes = [C.c_void_p, C.c_uint32]
    h = hostlib.bft_hosttest_build(km.ctypes.data, len(km), k, 0, 0)
    assert h
    try:
        hostlib.bft_hosttest_flatten(h, flat_min)
        for name in ("ccx", "f18", "fent"):
            n = C.c_uint64()
            assert hostlib.bft_hosttest_get_array(h, name.encode(), None, 0, C.byref(n)) == 0
            host = np.zeros(n.value, np.uint8)
            assert hostlib.bft_hosttest_get_array(h, name.encode(), host.ctypes.data, n.value, C.byref(n)) == 0
            dev = t.debug_array(name)
            assert dev.shape == host.shape, (name, dev.shape, host.shape)
            assert (dev == host).all(), name
            if flat_min == 1 and name != "ccx":
                assert n.value > 0
            if flat_min == 65536 and name != "ccx":
                assert n.value == 0
    finally:
        hostlib.bft_hosttest_free(h)


@pytest.mark.parametrize("k,ngen", [(27, 3), (27, 130), (27, 600), (31, 2), (31, 4), (31, 300), (18, 40)])
def test_composite_sort_builds_the_same_image(k, ngen):
    """One-word keys with ascending genome ids are sorted as (T << bits | genome) composites when that fits 63 bits
    ("build_composite" 1, the default) -- root-prefix buckets first, then every bucket on its own ("build_msd" 2 forces that at this
    small size; 0 = one device-wide sort) -- and by the general key + value sort otherwise: every array of the image, the colour
    sets and the extraction are identical -- including an incremental build and duplicate (k-mer, genome) pairs.  Where the key leaves
    7 bits or more (k <= 28) the insertion log itself holds the composites ("composite_log" 1, the default; 0 = k-mers + ids): 600 genomes
    at k = 27 pass the 512 ids a composite has room for, so that log is turned back into k-mers + ids half way."""
    from bloomfiltertrie_amd import BFT
    base = S.distinct(S.kmers_of(S.random_genome(30000, 5 + k), k))
    rng = np.random.default_rng(k + ngen)
    parts = [np.ascontiguousarray(base[rng.random(len(base)) < 0.3]) for _ in range(ngen)]
    imgs = []
    for comp, msd, clog in ((1, 2, 1), (0, 0, 1), (1, 0, 1), (1, 2, 0)):
        t = BFT(k)
        t.set_option("composite_log", clog)
        t.set_option("build_composite", comp)
        t.set_option("build_msd", msd)
        half = ngen // 2
        for g in range(half):
            t.insert_kmers(parts[g], g)
            if g == 0:
                t.insert_kmers(parts[g][::2], g)  # duplicates of the same pairs
        t.build()
        for g in range(half, ngen):  # second batch: the sorted store is merged with the new log
            t.insert_kmers(parts[g], g)
        t.build()
        ek, ecs = t.extract()
        imgs.append(({name: t.debug_array(name) for name in ARRAYS}, ek, ecs, [t.colorset(c) for c in sorted(set(ecs.tolist()))[:200]]))
        t.close()
    a = imgs[0]
    for b in imgs[1:]:
        for name in ARRAYS:
            assert (a[0][name] == b[0][name]).all(), name
        assert (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[3] == b[3]


def test_stream_ordered_inserts_build_the_same_image():
    """bft_gpu_insert_kmers_dev_async: batches converted on the caller's stream without a host round trip per call (their buffers
    released to the caching allocator right away, the log growing by doubling in between) give the image of the synchronised calls."""
    import torch
    from bloomfiltertrie_amd import BFT
    k, ngen = 27, 24
    dev = torch.device("cuda", 0)
    base = S.distinct(S.kmers_of(S.random_genome(120000, 17), k))
    rng = np.random.default_rng(4)
    parts = [np.ascontiguousarray(base[rng.random(len(base)) < 0.4]) for _ in range(ngen)]
    imgs = []
    for mode in ("sync", "async"):
        t = BFT(k)
        side = torch.cuda.Stream(device=dev)
        for g, part in enumerate(parts):
            if mode == "sync":
                d = torch.from_numpy(part).to(dev)
                torch.cuda.synchronize()
                t.insert_kmers_dev(d.data_ptr(), len(part), g)
            else:
                with torch.cuda.stream(side):
                    d = torch.from_numpy(part).to(dev, non_blocking=False)
                    t.insert_kmers_dev_async(d.data_ptr(), len(part), g, side.cuda_stream)
                    d.record_stream(side)
                    del d
                    junk = torch.randint(0, 255, (len(part) * 7,), dtype=torch.uint8, device=dev)  # reuses the freed block, stream-ordered
                    del junk
        t.build()
        ek, ecs = t.extract()
        imgs.append(({name: t.debug_array(name) for name in ARRAYS}, ek, ecs))
        t.close()
    for name in ARRAYS:
        assert (imgs[0][0][name] == imgs[1][0][name]).all(), name
    assert (imgs[0][1] == imgs[1][1]).all() and (imgs[0][2] == imgs[1][2]).all()
    # the null stream is a stream too (torch's default): same image again
    t = BFT(k)
    for g, part in enumerate(parts):
        d = torch.from_numpy(part).to(dev)
        t.insert_kmers_dev_async(d.data_ptr(), len(part), g, None)
        del d
    t.build()
    for name in ARRAYS:
        assert (imgs[0][0][name] == t.debug_array(name)).all(), name


def test_failed_build_leaves_the_previous_image_and_the_pending_insertions_intact():
    """bft_gpu_build is all-or-nothing: a failure after sorting, colour interning and assembly -- injected right
    before the commit point -- must leave the old image answering as before and the pending k-mers in the log; the next
    build then succeeds with everything."""
    from bloomfiltertrie_amd._lib import BFTError
    k = 27
    anc = S.random_genome(60000, 5)
    g0 = S.distinct(S.kmers_of(anc, k))
    g1 = S.distinct(S.kmers_of(S.mutate(anc, 0.03, 6), k))
    t = BFT(k)
    t.insert_kmers(g0, 0)
    t.build()
    q = np.concatenate([g0[::3], g1[::3], S.snp_mutants(g0[::7], k, 2)])
    before_bits, before_off, before_ids = t.query_colors(q)
    info0 = t.info()
    t.insert_kmers(g1, 1)
    assert t.info()["pending_pairs"] == len(g1)
    t.set_option("inject_build_failure", 1)
    with pytest.raises(BFTError, match="injected"):
        t.build()
    info1 = t.info()
    assert info1["pending_pairs"] == len(g1) and info1["kmers"] == info0["kmers"] and info1["pairs"] == info0["pairs"]
    t.build()  # the hook is one-shot: this build commits old + pending
    allk = S.distinct(np.concatenate([g0, g1]))
    info2 = t.info()
    assert info2["kmers"] == len(allk) and info2["pairs"] == len(g0) + len(g1) and info2["pending_pairs"] == 0
    bits, off, ids = t.query_colors(q)
    m0, m1 = S.member(q, g0), S.member(q, g1)
    assert (S.from_bits(bits, len(q)).astype(bool) == (m0 | m1)).all()
    for i in range(0, len(q), 97):
        assert ids[int(off[i]):int(off[i + 1])].tolist() == [g for g, m in ((0, m0[i]), (1, m1[i])) if m]
    # and the answers of the old image were those of genome 0 alone
    assert (S.from_bits(before_bits, len(q)).astype(bool) == m0).all() and len(before_ids) == int(m0.sum()) and before_off[-1] == len(before_ids)


def test_rejects_out_of_range_genome_ids():
    from bloomfiltertrie_amd._lib import BFTError
    t = BFT(27)
    km = S.distinct(S.kmers_of(S.random_genome(2000, 1), 27))
    for bad in (0xFFFFFFFF, 1 << 24):
        with pytest.raises(BFTError, match="id_genome"):
            t.insert_kmers(km, bad)
    t.insert_kmers(km, (1 << 24) - 1)  # the largest id accepted
    assert t.info()["pending_pairs"] == len(km)


def _colour_map(t):
    """{packed k-mer bytes: tuple of genome ids} of everything the handle stores"""
    km, cs = t.extract()
    sets = {c: tuple(t.colorset(c)) for c in np.unique(cs).tolist()}
    return {km[i].tobytes(): sets[int(cs[i])] for i in range(len(km))}, len(sets)


@pytest.mark.parametrize("k,flush", [(27, 4096), (27, 60000), (31, 20000), (45, 30000), (18, 5000)])
def test_insertions_merge_into_the_index_without_a_pair_bound(k, flush):
    """The index is its own store: a build sorts only what was inserted since the last one and merges that run into the index
    (bft_merge.hip).  With the flush threshold lowered ("flush_pairs": the log is merged before it holds that many pairs; 2^30 in
    production -- the reference inserts without bound, src/insertNode.c:18-36) dozens of merges happen during one series of insert
    calls: genome ids out of order, genomes inserted twice, a batch larger than the threshold (inserted in pieces).  The result --
    k-mers, colour set of every k-mer, the number of distinct colour sets and of (k-mer, genome) pairs, presence answers -- equals
    that of ONE build over everything, and ground truth."""
    rng = np.random.default_rng(k + flush)
    anc = S.random_genome(60000, k)
    genomes = [S.distinct(S.kmers_of(S.mutate(anc, 0.02, 10 + g), k)) for g in range(9)]
    order = [3, 0, 1, 2, 2, 5, 4, 8, 7, 6, 0]  # out of order, 2 and 0 twice
    whole, merged = BFT(k), BFT(k)
    merged.set_option("flush_pairs", flush)
    truth = {}
    for g in order:
        km = genomes[g]
        whole.insert_kmers(km, g)
        for part in np.array_split(km, 3):  # several insert calls per genome; some cross the threshold, one (flush 4096) exceeds it
            merged.insert_kmers(np.ascontiguousarray(part), g)
        for row in km:
            truth.setdefault(row.tobytes(), set()).add(g)
    whole.build()
    merged.build()
    mw, nsw = _colour_map(whole)
    mm, nsm = _colour_map(merged)
    assert mw == mm and nsw == nsm
    assert {kk: tuple(sorted(v)) for kk, v in truth.items()} == mm
    iw, im_ = whole.info(), merged.info()
    for f in ("kmers", "pairs", "colorsets", "nodes", "ccs", "prefixes", "child_nodes"):
        assert iw[f] == im_[f], f
    assert im_["pairs"] == sum(len(v) for v in truth.values()) and im_["pending_pairs"] == 0
    assert merged.footprint()["pair_store"] == 0
    allk = S.distinct(np.concatenate(genomes))
    q = np.concatenate([allk, S.snp_mutants(allk[::3], k, 4), S.pack_codes(rng.integers(0, 4, (3000, k), dtype=np.uint8))])
    assert (merged.query_presence(q) == whole.query_presence(q)).all()
    assert (S.from_bits(merged.query_presence(q), len(q)) == S.member(q, allk)).all()
    for arr in ARRAYS:  # the containers are a function of the k-mer set alone: bit-identical however the k-mers arrived
        assert (merged.debug_array(arr) == whole.debug_array(arr)).all(), arr
    with pytest.raises(Exception):
        merged.set_option("flush_pairs", 10)
    whole.close()
    merged.close()




def test_colour_interning_falls_back_to_exact_comparison_on_signature_collisions():
    """Colour sets are interned by a 64-bit signature; every k-mer's list is then compared with the dictionary entry it was given, and a
    mismatch makes the pass run again with the lists themselves compared.  With the test hook that makes a list's signature its LENGTH
    (different sets collide everywhere) the fallback must run and the result must be what the ordinary build gives."""
    k = 27
    anc = S.random_genome(40000, 3)
    genomes = [S.distinct(S.kmers_of(S.mutate(anc, 0.03, 40 + g), k)) for g in range(6)]
    a, b = BFT(k), BFT(k)
    b.set_option("test_weak_signature", 1)
    try:
        for g, km in enumerate(genomes):
            a.insert_kmers(km, g)
            b.insert_kmers(km, g)
        a.build()
        before = b.build_time()["intern_exact_passes"]
        b.build()
        assert b.build_time()["intern_exact_passes"] > before
    finally:
        b.set_option("test_weak_signature", 0)
    ma, na = _colour_map(a)
    mb, nb_ = _colour_map(b)
    assert ma == mb and na == nb_ and a.info()["colorsets"] == b.info()["colorsets"]
    a.close()
    b.close()


def test_dictionary_id_width_follows_the_largest_genome_id():
    """The genome ids of the colour-set dictionary are resident in 1, 2 or 4 bytes, whichever holds the largest id inserted so far;
    merges widen them on the way.  Colour ids, colour rows, the per-read tallies and a packed copy answer the same through every
    width, and ground truth."""
    import torch
    k = 27
    anc = S.random_genome(30000, 3)
    gids = [0, 1, 2, 255, 256, 4000, 65535, 65536, 70001]
    genomes = {g: S.distinct(S.kmers_of(S.mutate(anc, 0.02, 50 + i), k)) for i, g in enumerate(gids)}
    t = BFT(k)
    truth = {}
    widths = []
    for g in gids:
        t.insert_kmers(genomes[g], g)
        for row in genomes[g]:
            truth.setdefault(row.tobytes(), []).append(g)
        t.build()
        fp, info = t.footprint(), t.info()
        n_ids = sum(len(t.colorset(c)) for c in range(info["colorsets"]))
        widths.append((fp["colorset_dictionary"] - 4 * (info["colorsets"] + 1)) // max(n_ids, 1))
        q = genomes[g][::7]
        bits, off, ids = t.query_colors(q)
        assert S.from_bits(bits, len(q)).all()
        for i in range(0, len(q), 41):
            assert ids[int(off[i]):int(off[i + 1])].tolist() == truth[q[i].tobytes()]
    assert widths == [1, 1, 1, 1, 2, 2, 2, 4, 4]
    q = np.concatenate([genomes[g][::29] for g in gids])[:600]
    bits, rows = t.query_color_rows(q)
    for i in range(0, len(q), 13):
        assert np.flatnonzero(np.unpackbits(rows[i], bitorder="little")).tolist() == truth[q[i].tobytes()]
    reads = [bytes(S._ASCII[S.mutate(anc, 0.02, 50 + i)[100:400]]).decode() for i in (0, 4, 8)]
    got = t.query_sequences(reads, 0.9)
    assert [gids[0] in got[0], gids[4] in got[1], gids[8] in got[2]] == [True, True, True]
    n = t.image_size()
    blob = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    t.image_pack(blob.data_ptr(), n)
    b = BFT.from_image(blob.data_ptr(), n, device=0)
    assert b.footprint()["colorset_dictionary"] == t.footprint()["colorset_dictionary"]
    assert b.query_sequences(reads, 0.9) == got
    assert (b.query_color_rows(q)[1] == rows).all()
    mt, _ = _colour_map(t)
    assert mt == {kk: tuple(v) for kk, v in truth.items()}
    t.close()
    b.close()


def test_interning_when_nearly_every_colour_set_is_distinct():
    """The interning's hash table starts small (a pan-genome has few distinct lists) and is retried at twice the number of k-mers when
    it fills beyond half: 60 000 k-mers with random subsets of 40 genomes carry ~60 000 distinct sets, more than half of the first
    table.  Sets, their number and every k-mer's set equal ground truth."""
    k = 27
    base = S.distinct(S.kmers_of(S.random_genome(60000 + k - 1, 77), k))
    rng = np.random.default_rng(5)
    member = rng.random((40, len(base))) < 0.5
    member[0, ~member.any(axis=0)] = True  # every k-mer is in some genome
    t = BFT(k)
    before = t.build_time()["intern_exact_passes"]
    for g in range(40):
        t.insert_kmers(np.ascontiguousarray(base[member[g]]), g)
    t.build()
    assert t.build_time()["intern_exact_passes"] == before  # (the retry is not the exact fallback)
    truth = {base[i].tobytes(): tuple(np.flatnonzero(member[:, i]).tolist()) for i in range(len(base))}
    got, nsets = _colour_map(t)
    assert got == truth
    assert nsets == len(set(truth.values())) == t.info()["colorsets"] and nsets > 40000
    t.close()


@pytest.mark.parametrize("k,gids", [(27, [0, 1, 2, 3, 4]), (27, [0, 1]), (31, [0, 1, 2, 3, 4]), (31, [3, 7]), (31, [0, 300, 301, 70000]), (27, [5, 600, 601, 602, 603]), (20, [0, 1, 2]),
                                    (9, [0, 1, 2])])
def test_bucket_sort_rank_modes_build_the_same_image(k, gids):
    """A root-prefix bucket ranks its digits with one LDS atomic per key, checks the final order over every bit -- k-mers ascending,
    the ids of a k-mer in insertion order -- and goes on a list when the check fails: a second launch sorts the listed buckets again with
    ballot ranks (stable by construction; bft_front.hip: a wavefront per bucket up to 1024 composites, a workgroup per larger one, one
    fallback kernel).  "test_front_rank_mode": 0 as shipped, 1 ballots only, 2 the check always fails (every bucket is sorted twice).  Image, extraction and colour sets are those of the
    device-wide sort in every mode -- composites that fit 63 bits and those that only fit inside a bucket (k = 31; ids of one, two
    and four bytes) --; as shipped no bucket fails the check."""
    n_pref = 40
    base = S.low_entropy_kmers(30000, k, n_pref, 11 + k) if k >= 18 else S.distinct(S.pack_codes(np.random.default_rng(k).integers(0, 4, (30000, k), dtype=np.uint8)))
    rng = np.random.default_rng(k + len(gids))
    parts = [np.ascontiguousarray(base[rng.random(len(base)) < 0.5]) for _ in gids]  # ~375 k-mers per genome and bucket
    imgs, redone = [], []
    try:
        for msd, mode in ((0, 0), (2, 0), (2, 1), (2, 2)):
            t = BFT(k)
            t.set_option("build_msd", msd)
            t.set_option("test_front_rank_mode", mode)
            for i, (g, p) in enumerate(zip(gids, parts)):
                t.insert_kmers(p, g)
                if i == 1:
                    t.insert_kmers(p[::3], g)  # duplicate pairs
            t.build()
            bt = t.build_time()
            redone.append(int(bt["sort_redone_buckets"]))
            assert (bt["sort_max_bucket"] > 0) == (msd == 2)
            assert bt["sort_max_bucket"] <= 4096
            ek, ecs = t.extract()
            imgs.append(({name: t.debug_array(name) for name in ARRAYS}, ek, ecs, [list(t.colorset(c)) for c in sorted(set(ecs.tolist()))[:100]]))
            t.close()
    finally:
        w = BFT(k)
        w.set_option("test_front_rank_mode", 0)
        w.close()
    a = imgs[0]
    for b in imgs[1:]:
        for name in ARRAYS:
            assert (a[0][name] == b[0][name]).all(), name
        assert (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[3] == b[3]
    assert redone[0] == 0 and redone[1] == 0, redone  # as shipped no bucket fails the check
    # (the ballot ranks are a launch of their own over a list of buckets: mode 1 puts every bucket on it unsorted, mode 2 after a sort whose check
    # is made to fail; k = 9: the split covers every bit, nothing is sorted in a bucket)
    assert (redone[2] >= n_pref and redone[3] >= n_pref) if k > 9 else (redone[2] == 0 and redone[3] == 0), redone


def test_k32_builds_through_the_device_wide_sort_when_buckets_are_forced():
    """k = 32: the root-prefix split of (k-mer, id) pairs would sort the bit range [46, 64), which rocPRIM's radix sort mis-sorts (ranges that
    start above bit 0 and end at bit 64): "build_msd" 2 must not take it.  Found by tools/stress_parity.py (wrong presence bits, then a
    memory fault in the bucket kernels).  Same image, extraction and answers as the device-wide sort."""
    k = 32
    base = S.low_entropy_kmers(12000, k, 12, 17, levels=1)
    rng = np.random.default_rng(5)
    parts = [np.ascontiguousarray(base[rng.random(len(base)) < 0.6]) for _ in range(4)]
    q = np.ascontiguousarray(np.concatenate([base, S.snp_mutants(base[::3], k, 3)]))
    truth = np.zeros(len(q), bool)
    for p in parts:
        truth |= S.member(q, p)
    outs = []
    for msd in (0, 2):
        t = BFT(k)
        t.set_option("build_msd", msd)
        for g, p in enumerate(parts):
            t.insert_kmers(p, g)
        t.build()
        assert t.build_time()["sort_max_bucket"] == 0  # (no bucket was sorted: the split was not taken)
        bits = S.from_bits(t.query_presence(q), len(q)).astype(bool)
        assert (bits == truth).all(), msd
        outs.append(t.extract())
        t.close()
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()


@pytest.mark.parametrize("k", [27, 31, 18, 45, 63])
def test_compact_table_drops_and_restores_the_sorted_table(k, tmp_path):
    """"compact_table": once the k-mer hash holds every (k-mer, colour set) the sorted table and the colour set per k-mer leave HBM;
    presence, colour-row, branching and sequence queries go on as before; rows, extraction, a .bft file, a packed image and a merge of
    new insertions bring the table back first.  Everything equals what a handle without the option gives."""
    import torch
    anc = S.random_genome(50000, k)
    genomes = [S.distinct(S.kmers_of(S.mutate(anc, 0.02, 20 + g), k)) for g in range(6)]
    a, b = BFT(k), BFT(k)
    a.set_option("compact_table", 0)  # (the option is on by default since round 4)
    b.set_option("compact_table", 1)
    for t in (a, b):
        for g in range(4):
            t.add_genome(f"g{g}.fa")
            t.insert_kmers(genomes[g], g)
        t.build()
    fa, fb = a.footprint(), b.footprint()
    assert fb["kmer_table"] == 0 and fb["colorset_per_kmer"] == 0 and fa["kmer_table"] > 0
    assert b.info()["image_bytes"] < a.info()["image_bytes"] - 12 * a.info()["kmers"] + 1024
    allk = S.distinct(np.concatenate(genomes[:4]))
    q = np.concatenate([allk[::3], S.snp_mutants(allk[::7], k, 4)])
    assert (a.query_presence(q) == b.query_presence(q)).all()

    def rows_dev(t):  # the device-resident colour rows (the host call goes through row numbers)
        dq = torch.from_numpy(q).to("cuda:0")
        bits = torch.zeros(((len(q) + 63) // 64) * 8, dtype=torch.uint8, device="cuda:0")
        rows = torch.zeros((len(q), (t.info()["genomes"] + 7) // 8), dtype=torch.uint8, device="cuda:0")
        scratch = torch.zeros(len(q), dtype=torch.int32, device="cuda:0")
        t.query_color_rows_dev(dq.data_ptr(), len(q), bits.data_ptr(), rows.data_ptr(), scratch.data_ptr())
        torch.cuda.synchronize()
        return bits.cpu().numpy(), rows.cpu().numpy()

    (ba, wa), (bb, wb) = rows_dev(a), rows_dev(b)
    assert (ba == bb).all() and (wa == wb).all()
    assert (a.query_branching(q[:4000]) == b.query_branching(q[:4000])).all()
    reads = [bytes(S._ASCII[S.mutate(anc, 0.02, 20 + g)[500:700]]).decode() for g in range(4)]
    assert a.query_sequences(reads, 0.9) == b.query_sequences(reads, 0.9)
    assert b.footprint()["kmer_table"] == 0  # still away: none of those needed it
    # rows / colour ids / extraction bring it back
    ra, rb = a.query_color_rows(q), b.query_color_rows(q)
    assert (ra[0] == rb[0]).all() and (ra[1] == rb[1]).all() and (ra[1] == wa).all()
    b.set_option("compact_table", 1)
    ca, cb = a.query_colors(q[:3000]), b.query_colors(q[:3000])
    assert all((x == y).all() for x, y in zip(ca, cb))
    ka, sa = a.extract()
    kb, sb = b.extract()
    assert (ka == kb).all() and (sa == sb).all()
    assert b.footprint()["kmer_table"] == fa["kmer_table"]
    for name in ARRAYS:
        assert (a.debug_array(name) == b.debug_array(name)).all(), name
    b.set_option("compact_table", 1)  # away again
    assert b.footprint()["kmer_table"] == 0
    if k % 9 == 0:  # (the .bft format's k)
        a.write_bft(str(tmp_path / "a.bft"))
        b.write_bft(str(tmp_path / "b.bft"))
        assert open(tmp_path / "a.bft", "rb").read() == open(tmp_path / "b.bft", "rb").read()
        assert b.footprint()["kmer_table"] > 0
        b.set_option("compact_table", 1)
    n = b.image_size()
    blob = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    b.image_pack(blob.data_ptr(), n)
    c = BFT.from_image(blob.data_ptr(), n, device=0)
    assert (c.query_presence(q) == a.query_presence(q)).all() and (c.extract()[0] == ka).all()
    c.close()
    # a merge of new insertions
    b.set_option("compact_table", 1)
    for t in (a, b):
        for g in (4, 5):
            t.add_genome(f"g{g}.fa")
            t.insert_kmers(genomes[g], g)
        t.build()
    assert b.footprint()["kmer_table"] == 0
    q2 = np.concatenate([genomes[5][::2], q[:5000]])
    assert (a.query_presence(q2) == b.query_presence(q2)).all()
    ra, rb = a.query_color_rows(q2), b.query_color_rows(q2)
    assert (ra[1] == rb[1]).all()
    ma, na = _colour_map(a)
    mb, nb_ = _colour_map(b)
    assert ma == mb and na == nb_
    # the k-mer hash switched off: the walk needs the table
    b.set_option("compact_table", 1)
    b.set_option("kmer_hash", 0)
    assert b.footprint()["kmer_table"] > 0
    assert (a.query_presence(q2) == b.query_presence(q2)).all()
    # options that re-derive tables FROM the sorted table while it is away bring it back themselves (round 3's advice: "flat_min" and
    # "root_direct" read a NULL table and refilled the k-mer hash from it)
    b.set_option("kmer_hash", 1)
    b.set_option("compact_table", 1)
    assert b.footprint()["kmer_table"] == 0
    for name, v in (("flat_min", 2000), ("root_direct", 1), ("root_direct", 3), ("root_quartiles", 0), ("root_quartiles", 1), ("walk_hash", 1)):
        b.set_option(name, v)
        assert b.footprint()["kmer_table"] == 0 or name == "walk_hash", name
        assert (a.query_presence(q2) == b.query_presence(q2)).all(), name
    ra, rb = a.query_color_rows(q2), b.query_color_rows(q2)
    assert (ra[0] == rb[0]).all() and (ra[1] == rb[1]).all()
    a.close()
    b.close()


def test_host_inserts_through_the_pinned_ring_keep_their_order_and_content():
    """Host batches up to a megabyte are copied into a ring of eight pinned slots that the packing kernel reads asynchronously
    (bft_gpu_insert_kmers); larger ones take the staged copy.  Forty small batches (the ring wraps five times), a large one in the
    middle and the caller's buffer overwritten right after every call: k-mers, colour sets and answers equal ground truth."""
    k = 27
    rng = np.random.default_rng(11)
    anc = S.random_genome(400000, 9)
    big = S.distinct(S.kmers_of(anc, k))  # ~4x10^5 k-mers x 7 bytes: beyond a slot
    assert big.nbytes > (1 << 20)
    t = BFT(k)
    truth = {}
    scratch = np.zeros((30000, big.shape[1]), dtype=np.uint8)
    for g in range(41):
        if g == 20:
            t.insert_kmers(big, g)
            src = big
        else:
            src = np.ascontiguousarray(big[rng.choice(len(big), int(rng.integers(1, 30000)), replace=False)])
            scratch[: len(src)] = src
            t.insert_kmers(scratch[: len(src)], g)
            scratch[:] = 0xFF  # the library must have taken its copy
        for row in src:
            truth.setdefault(row.tobytes(), []).append(g)
    t.build()
    got, _ = _colour_map(t)
    assert got == {kk: tuple(v) for kk, v in truth.items()}
    q = np.concatenate([big[::5], S.snp_mutants(big[::11], k, 2)])
    assert (S.from_bits(t.query_presence(q), len(q)) == S.member(q, big)).all()
    t.close()


@pytest.mark.parametrize("k,gids", [(27, list(range(12))), (31, list(range(12))), (31, [0, 400, 70000])])
def test_a_bucket_beyond_the_capacity_falls_back_to_the_device_wide_sort(k, gids):
    """The size of the largest root-prefix bucket reaches the host while the wavefront kernel is already sorting the smaller buckets in
    place (bft_front_buckets); when that bucket is beyond what a workgroup sorts (4096 composites) nothing is produced and the build
    takes the device-wide sort from the insertion log.  Three prefixes carry every k-mer here: image, extraction and colour sets equal
    those of a handle that never tried the buckets, and the handle reports the bucket it met."""
    base = S.low_entropy_kmers(9000, k, 3, 5 + k)  # ~3000 k-mers per prefix, times the genomes that hold them
    rng = np.random.default_rng(k)
    parts = [np.ascontiguousarray(base[rng.random(len(base)) < 0.6]) for _ in gids]
    imgs = []
    for msd in (0, 2):
        t = BFT(k)
        t.set_option("build_msd", msd)
        for g, p in zip(gids, parts):
            t.insert_kmers(p, g)
        t.build()
        bt = t.build_time()
        if msd == 2:
            assert bt["sort_max_bucket"] > 4096, bt["sort_max_bucket"]
        ek, ecs = t.extract()
        imgs.append(({name: t.debug_array(name) for name in ARRAYS}, ek, ecs, [list(t.colorset(c)) for c in sorted(set(ecs.tolist()))[:100]]))
        q = np.concatenate([base[::3], S.snp_mutants(base[::7], k, 1)])
        assert (S.from_bits(t.query_presence(q), len(q)) == S.member(q, np.concatenate(parts))).all()
        t.close()
    a, b = imgs
    for name in ARRAYS:
        assert (a[0][name] == b[0][name]).all(), name
    assert (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[3] == b[3]



def test_build_stages_table():
    """bft_gpu_build_stages ("build_stages" 1): the GPU time of the last build stage by stage -- HIP events on the build's stream, the algorithmic
    bytes of the streaming stages beside them; off by default (no table, no events)."""
    from bloomfiltertrie_amd import BFT
    k = 27
    anc = S.random_genome(300000, 21)
    t = BFT(k)
    t.insert_kmers(S.distinct(S.kmers_of(anc, k)), 0)
    t.build()
    assert t.build_stages() == []
    t.set_option("build_stages", 1)
    t.set_option("build_msd", 2)
    for g in range(1, 4):
        t.insert_kmers(S.distinct(S.kmers_of(S.mutate(anc, 0.01, g), k)), g)
    t.build()  # (a merge into the first index)
    st = t.build_stages()
    names = [n for n, _, _ in st]
    assert len(st) >= 10 and all(ms >= 0 for _, ms, _ in st) and sum(ms for n, ms, _ in st if not n.startswith("+")) > 0
    assert any("merge into the index" in n for n in names) and any(n.startswith("containers depth 0") for n in names) and any("root tables" in n for n in names)
    assert any(by > 0 for _, _, by in st)
    t.set_option("build_stages", 0)
    t.insert_kmers(S.distinct(S.kmers_of(S.mutate(anc, 0.01, 9), k)), 4)
    t.build()
    assert [n for n, _, _ in t.build_stages()] == names  # (the table of the last build that recorded one stays)
    t.close()


@pytest.mark.gpu
def test_closed_handles_leave_their_blocks_in_the_cache():
    """bft_gpu_free hands the handle's device blocks to the library's cache (the next handle's build of the same size takes them instead of paying
    hipMalloc: 1 ms to 0.4 s per gigabyte block depending on the box); bft_gpu_cache_release gives the cache back to the runtime."""
    import torch
    from bloomfiltertrie_amd import BFT, cache_release
    cache_release()
    k = 27
    km = S.distinct(S.kmers_of(S.random_genome(400000, 77), k))
    q = np.concatenate([km[:50000], S.snp_mutants(km[:50000], k, 5)])
    answers = []
    for rnd in range(2):
        t = BFT(k)
        t.insert_kmers(km, 0)
        t.insert_kmers(km[::3], 1)
        t.build()
        answers.append(S.from_bits(t.query_presence(q), len(q)))
        t.close()
    assert (answers[0] == answers[1]).all() and (answers[0] == S.member(q, km)).all()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    released = cache_release()
    assert released > len(km) * 8  # (the closed handles' table alone is that large)
    assert torch.cuda.mem_get_info()[0] >= free0 + released // 2
    assert cache_release() == 0
    # with a handle alive: the transients of its build are cached under its own stream; released, and the handle builds on
    t = BFT(k)
    t.insert_kmers(km, 0)
    t.build()
    assert cache_release() > 0
    t.insert_kmers(km[::3], 1)
    t.build()
    assert (S.from_bits(t.query_presence(q), len(q)) == answers[0]).all()
    _, off, ids = t.query_colors(km[:3])  # (k-mer 0 is in both genomes, 1 and 2 in the first only)
    assert off.tolist() == [0, 2, 3, 4] and ids[:4].tolist() == [0, 1, 0, 0]
    t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k,ngen,glen,rate", [(36, 40, 6000, 0.01), (45, 300, 700, 0.01), (63, 1200, 260, 0.01), (63, 24, 20000, 0.2), (64, 64, 3000, 0.02)])
def test_two_word_front_end_builds_the_same_image(k, ngen, glen, rate):
    """Two-word keys (33 <= k <= 64) with ascending genome ids: the root-prefix split on the top 18 T bits, then every bucket on its own in LDS
    over the two words (bft_front.hip: k_bucket2_tiny / k_bucket2_sort_wave / k_bucket2_sort by size; "build_msd" 2 forces it at this size) against
    the device-wide sort of every word ("build_msd" 0): every array of the image, the colour sets and the extraction are identical --
    buckets of one locus with hundreds of copies (many genomes of a short ancestor), buckets of a few all-distinct k-mers (few genomes, many
    SNPs), duplicate pairs, and an incremental build on top."""
    from bloomfiltertrie_amd import BFT
    anc = S.random_genome(glen, 1000 + k)
    parts = [S.distinct(S.kmers_of(S.mutate(anc, rate, 7 * k + g), k)) for g in range(ngen)]
    imgs = []
    for msd in (2, 0):
        t = BFT(k)
        t.set_option("build_msd", msd)
        half = ngen // 2
        for g in range(half):
            t.insert_kmers(parts[g], g)
            if g == 1:
                t.insert_kmers(parts[g][::3], g)  # duplicates of the same pairs
        t.build()
        if msd == 2:
            assert t.build_time()["sort_max_bucket"] > 0  # (the front end ran: it reports its largest bucket)
        for g in range(half, ngen):
            t.insert_kmers(parts[g], g)
        t.build()
        ek, ecs = t.extract()
        imgs.append(({name: t.debug_array(name) for name in ARRAYS}, ek, ecs, [t.colorset(c) for c in sorted(set(ecs.tolist()))[:200]], t.info()))
        t.close()
    a, b = imgs
    for name in ARRAYS:
        assert (a[0][name] == b[0][name]).all(), name
    assert (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[3] == b[3]
    assert a[4]["kmers"] == b[4]["kmers"] == len(S.distinct(np.concatenate(parts))) and a[4]["pairs"] == b[4]["pairs"] == sum(len(x) for x in parts)





@pytest.mark.parametrize("k,ngen,glen", [(27, 6, 400000), (63, 5, 300000), (99, 3, 200000)])
def test_ballot_ranks_in_the_device_wide_sort_build_the_same_image(k, ngen, glen):
    """bft_sort.h ranks the keys of a tile with one LDS atomic per key where the device serves such an instruction's lanes in lane order (checked once
    per process by a kernel) and with wavefront ballots otherwise; "sort_ballots" 1 forces the ballots -- the path a device that fails the check would
    take, in every sort of the build (root-prefix split, k-mer hash sort, the assembly's and the interning's sorts; one-word, two-word and longer keys).
    Same image, same extraction."""
    from bloomfiltertrie_amd import BFT
    anc = S.random_genome(glen, k)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 40 + g) if g else anc, k)) for g in range(ngen)]
    imgs = []
    try:
        for ballots in (0, 1):
            t = BFT(k)
            t.set_option("sort_ballots", ballots)
            t.set_option("build_msd", 2)
            for g, km in enumerate(gk):
                t.insert_kmers(km, g)
            t.build()
            ek, ecs = t.extract()
            imgs.append(({name: t.debug_array(name) for name in ARRAYS}, ek, ecs, t.info()))
            q = np.concatenate([gk[0][::5], S.snp_mutants(gk[1][::7], k, 3)])
            imgs[-1] += (t.query_presence(q), t.query_colors(q))
            t.close()
    finally:
        w = BFT(k)
        w.set_option("sort_ballots", 0)
        w.close()
    a, b = imgs
    for name in ARRAYS:
        assert (a[0][name] == b[0][name]).all(), name
    assert (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[3] == b[3]
    assert (a[4] == b[4]).all() and all((x == y).all() for x, y in zip(a[5], b[5]))
