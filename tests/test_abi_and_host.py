"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/bft_gpu.h declares, fails loudly
without a device, and the host logic that feeds the kernels (T-form conversion, container assembly, the shared
per-query walk of csrc/bft_walk.h) matches the oracle and ground truth."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from bloomfiltertrie_amd import _lib, synth as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def built():
    subprocess.check_call(["make", "-C", _lib.CSRC, "all"], stdout=subprocess.DEVNULL)
    return True


def test_header_symbols_are_exported(built):
    hdr = open(_lib.HEADER).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bft_gpu_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 18
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (bft_gpu_[a-z_0-9]+)", out))
    assert declared <= exported


def test_fails_loudly_without_a_device(built):
    lib = _lib.load()
    if lib.bft_gpu_device_count() > 0:
        pytest.skip("a GPU is present")
    from bloomfiltertrie_amd import BFT
    with pytest.raises(_lib.BFTError):
        BFT(27)


def test_rejects_k_out_of_range(built):
    lib = _lib.load()
    h = C.c_void_p()
    for k in (8, 127, 135, 0, -9):
        assert lib.bft_gpu_create(k, 0, C.byref(h)) == -1  # BFT_GPU_E_ARG, before any device is touched
        assert b"[9,126]" in lib.bft_gpu_last_error()


def test_product_does_not_reference_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "bloomfiltertrie_amd")):
        for f in fs:
            if f.endswith((".py", ".h", ".cpp", ".hip", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle|liborc|bft_oracle", txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


# ---- host logic through the test-only helper library ---------------------------------------------------------------
@pytest.fixture(scope="session")
def hostlib(built):
    lib = C.CDLL(os.path.join(_lib.CSRC, "libbft_hosttest.so"))
    lib.bft_hosttest_build.restype = C.c_void_p
    lib.bft_hosttest_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int]
    lib.bft_hosttest_query.restype = C.c_uint64
    lib.bft_hosttest_query.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.bft_hosttest_set_probe.argtypes = [C.c_void_p, C.c_int]
    lib.bft_hosttest_root_direct.argtypes = [C.c_void_p, C.c_int]
    lib.bft_hosttest_kmer_hash.restype = C.c_uint64
    lib.bft_hosttest_kmer_hash.argtypes = [C.c_void_p, C.c_uint32]
    lib.bft_hosttest_query_kh.restype = C.c_int64
    lib.bft_hosttest_query_kh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.bft_hosttest_kh_geometry.argtypes = [C.c_void_p, C.c_void_p]
    lib.bft_hosttest_walk_kh.argtypes = [C.c_void_p, C.c_int]
    lib.bft_hosttest_kh_roundtrip.argtypes = [C.c_void_p]
    lib.bft_hosttest_kh_homes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.bft_hosttest_kh_probe_stats.restype = C.c_double
    lib.bft_hosttest_kh_probe_stats.argtypes = [C.c_void_p, C.c_void_p]
    lib.bft_hosttest_node_hash.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.bft_hosttest_stats.argtypes = [C.c_void_p, C.c_void_p]
    lib.bft_hosttest_roundtrip.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    lib.bft_hosttest_hashmod.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.bft_hosttest_free.argtypes = [C.c_void_p]
    return lib


def test_hash_table_matches_oracle_and_reference(hostlib, oracle_mod):
    hm = np.zeros(16384, np.uint32)
    hostlib.bft_hosttest_hashmod(oracle_mod._load().orc_create and 1804289383, 846930886, hm.ctypes.data)
    t = oracle_mod.OracleBFT(27)
    hv = t.hash_v(2 * 16384)
    exp = np.array([(hv[2 * i] % 1504) | ((hv[2 * i + 1] % 1504) << 16) for i in range(16384)], dtype=np.uint32)
    assert (hm == exp).all()
    ref = oracle_mod.ref_prims()
    if ref is not None:  # the reference's own xxhash.c
        for i in (0, 1, 2, 77, 16383):
            key = bytes([(i >> 10) & 0xFF, (i >> 2) & 0xFF, (i << 6) & 0xFF])
            assert (hm[i] & 0xFFFF) == ref.BFT_HASH_XXH64(key, 3, 1804289383) % 1504
            assert (hm[i] >> 16) == ref.BFT_HASH_XXH64(key, 3, 846930886) % 1504


@pytest.mark.parametrize("k", [9, 18, 27, 36, 45, 54, 63, 72, 81, 90, 99, 108, 117, 126])
def test_tform_roundtrip(hostlib, k):
    km = np.ascontiguousarray(S.pack_codes(np.random.default_rng(k).integers(0, 4, (2000, k), dtype=np.uint8)))
    out = np.zeros_like(km)
    W = (2 * k + 63) // 64
    tf = np.zeros((len(km), W), dtype=np.uint64)
    hostlib.bft_hosttest_roundtrip(km.ctypes.data, len(km), k, out.ctypes.data, tf.ctypes.data)
    assert (out == km).all()
    # T-form order == order of the per-level rotated prefixes n2..n9,n1 (src/presenceNode.c:1367-1371)
    codes = S.unpack_codes(km, k)
    for i in range(0, 50):
        val = 0
        for d in range(k // 9):
            n = codes[i, 9 * d:9 * d + 9].tolist()
            r = 0
            for x in n[1:] + n[:1]:
                r = (r << 2) | x
            val = (val << 18) | r
        got = 0
        for w in range(W):
            got = (got << 64) | int(tf[i, w])
        assert got == val


def _host_check(hostlib, oracle_mod, km, k, seed=0):
    km = np.ascontiguousarray(km)
    h = hostlib.bft_hosttest_build(km.ctypes.data, len(km), k, 0, 0)
    assert h
    rng = np.random.default_rng(seed)
    parts = [km, S.pack_codes(rng.integers(0, 4, (max(500, len(km) // 2), k), dtype=np.uint8))]
    if len(km):
        parts.append(S.snp_mutants(km, k, seed + 1))
    q = np.concatenate(parts)
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    bits = np.zeros((len(q) + 7) // 8, np.uint8)
    rows = np.zeros(len(q), np.uint32)
    hostlib.bft_hosttest_query(h, q.ctypes.data, len(q), bits.ctypes.data, rows.ctypes.data)
    bits8, rows8 = np.zeros_like(bits), np.zeros_like(rows)
    hostlib.bft_hosttest_set_probe(h, 1)  # 8-row blocks + re-interpolated guesses: a tuning mode, same answers
    hostlib.bft_hosttest_query(h, q.ctypes.data, len(q), bits8.ctypes.data, rows8.ctypes.data)
    hostlib.bft_hosttest_set_probe(h, 0)
    assert (bits8 == bits).all() and (rows8 == rows).all()
    # root level through the derived direct table (1), through range + direct tables (2), and with the quartile table that steers the
    # search of a plain group (3; in every probe mode): same answers, same rows
    for rd, mode in ((1, 0), (2, 0), (3, 0), (3, 1)):
        hostlib.bft_hosttest_root_direct(h, rd)
        hostlib.bft_hosttest_set_probe(h, mode)
        hostlib.bft_hosttest_query(h, q.ctypes.data, len(q), bits8.ctypes.data, rows8.ctypes.data)
        hostlib.bft_hosttest_root_direct(h, 0)
        hostlib.bft_hosttest_set_probe(h, 0)
        assert (bits8 == bits).all() and (rows8 == rows).all(), (rd, mode)
    # the k-mer hash (BFT_KH_*; the host build stores the row as the value): the lookup the kernels run gives the walk's answers at the
    # default occupancy, at a sparse one and at 80 % (runs of full lines), for every k (1 to 10 slots per line); every slot decodes back to
    # its k-mer (what "compact_table" rebuilds the sorted table from)
    for load in (60, 10, 80):
        nl = hostlib.bft_hosttest_kmer_hash(h, load)
        if len(km) == 0:
            assert nl == 0 and hostlib.bft_hosttest_query_kh(h, q.ctypes.data, len(q), bits8.ctypes.data, rows8.ctypes.data) == -1
            continue
        geo = np.zeros(14, np.uint32)
        hostlib.bft_hosttest_kh_geometry(h, geo.ctypes.data)
        S_, f_, wb_, cb_, kb_, qb_, hb_, restb_, t_, m_, nl_, maxd_, db_, novf_ = (int(x) for x in geo)
        n_st = len(S.distinct(km))
        assert nl > 0
        # (hashed: the top min(32, 2k - 4) bits without the first nucleotide's two -- those lie below them only for k < 11)
        assert hb_ == min(32, 2 * k - 4) - (2 if k >= 11 else 0) and restb_ == 2 * k - hb_ and (1 << cb_) > n_st and f_ == min(32, 128 // S_ - 1) and kb_ == max(restb_ + qb_, f_)
        assert wb_ == 48 // S_ and cb_ + db_ + kb_ - f_ <= 8 * wb_ and nl == nl_ == (m_ << (hb_ - t_)) and t_ <= 27 and 2 <= m_ <= 32
        assert db_ == (3 if S_ >= 6 else 4 if S_ >= 4 else 5 if S_ == 3 else 6 if S_ == 2 else 8) and maxd_ < (1 << db_)
        assert nl * S_ * load >= n_st * 100 and (1 << qb_) >= -(-(1 << t_) // m_) and novf_ <= (0 if load <= 60 else 4096)
        got = hostlib.bft_hosttest_query_kh(h, q.ctypes.data, len(q), bits8.ctypes.data, rows8.ctypes.data)
        assert got == int(S.from_bits(bits, len(q)).sum()) and (bits8 == bits).all() and (rows8 == rows).all(), load
        assert hostlib.bft_hosttest_kh_roundtrip(h) == 1, load
        # the four successors of a k-mer (its last k - 1 nucleotides + any fourth) share their home line, and so do its four predecessors:
        # what lets a branching query read two lines instead of eight (src/branchingNode.c:16-112, :240-340)
        if load == 60:
            fam = []
            for row in S.unpack_codes(q[:64], k):
                fam += [np.concatenate([row[1:], [c]]) for c in range(4)] + [np.concatenate([[c], row[:-1]]) for c in range(4)]
            fq = np.ascontiguousarray(S.pack_codes(np.array(fam, dtype=np.uint8)))
            homes = np.zeros(len(fq), np.uint64)
            assert hostlib.bft_hosttest_kh_homes(h, fq.ctypes.data, len(fq), homes.ctypes.data) == 0
            homes = homes.reshape(-1, 2, 4)
            assert (homes == homes[:, :, :1]).all(), k
            if nl > 256 and len(q) >= 64:
                assert len(np.unique(homes[:, :, 0])) > homes.shape[0]  # (the families do not all land on a few lines)
        # ... and the container walk that looks plain root groups up in the table (the table's values are the rows here)
        hostlib.bft_hosttest_root_direct(h, 2)
        if hostlib.bft_hosttest_walk_kh(h, 1):
            hostlib.bft_hosttest_query(h, q.ctypes.data, len(q), bits8.ctypes.data, rows8.ctypes.data)
            assert (bits8 == bits).all() and (rows8 == rows).all(), load
            hostlib.bft_hosttest_walk_kh(h, 0)
        hostlib.bft_hosttest_root_direct(h, 0)
        worst = C.c_uint64()
        mean = hostlib.bft_hosttest_kh_probe_stats(h, C.byref(worst))
        # (one slot per line at 80 %: the textbook (1 + 1 / (1 - a)) / 2 = 3 lines per successful lookup; eight slots at 60 %: 1.05)
        assert 1.0 <= mean < ((1.25 if S_ >= 5 else 2.0) if load <= 60 else 3.6), (load, S_, mean, worst.value)
    hostlib.bft_hosttest_kmer_hash(h, 0)
    o = oracle_mod.OracleBFT(k)
    o.insert_kmers(km, 0)
    assert (bits == o.query_presence(q)).all()
    assert (S.from_bits(bits, len(q)) == S.member(q, km)).all()
    for tiny in (0, 1):  # levels below the root through the node prefix hash (full-size table; 8-bucket table: nearly every bucket full)
        hostlib.bft_hosttest_node_hash(h, 1, tiny)
        hostlib.bft_hosttest_query(h, q.ctypes.data, len(q), bits8.ctypes.data, rows8.ctypes.data)
        hostlib.bft_hosttest_node_hash(h, 0, 0)
        assert (bits8 == bits).all() and (rows8 == rows).all()
    st = np.zeros(12, np.uint64)
    hostlib.bft_hosttest_stats(h, st.ctypes.data)
    hostlib.bft_hosttest_free(h)
    return st.tolist()


@pytest.mark.parametrize("k", [9, 18, 27, 36, 63, 126])
def test_host_index_random(hostlib, oracle_mod, k):
    st = _host_check(hostlib, oracle_mod, S.distinct(S.kmers_of(S.random_genome(80000, 7 + k), k)), k)
    assert st[2] > 1  # several CCs in the root


@pytest.mark.parametrize("k,levels", [(18, 1), (27, 2), (36, 3), (63, 3), (45, 4)])
def test_host_index_deep(hostlib, oracle_mod, k, levels):
    st = _host_check(hostlib, oracle_mod, S.low_entropy_kmers(60000, k, 24, seed=k + levels, levels=levels), k)
    assert st[4] > 0  # child nodes


@pytest.mark.parametrize("n", [0, 1, 254, 255, 256, 300])
def test_host_index_tiny(hostlib, oracle_mod, n):
    km = S.distinct(S.pack_codes(np.random.default_rng(n).integers(0, 4, (n, 27), dtype=np.uint8))) if n else np.zeros((0, 7), np.uint8)
    st = _host_check(hostlib, oracle_mod, km, 27)
    if len(km) < 255:
        assert st[2] == 0 and st[9] == len(km)  # UC only, as src/insertNode.c:183-223
    else:
        assert st[2] >= 1


def test_host_index_invariants_a7(hostlib, oracle_mod):
    """UC non-empty => last CC holds >= 255 prefixes; UC < 255 rows; both filter2 geometries."""
    km = S.distinct(S.kmers_of(S.random_genome(600000, 3), 27))
    st = _host_check(hostlib, oracle_mod, km, 27)
    assert st[6] > 0 and st[6] < st[2]
    assert st[9] < 255


# ---- extension beyond the reference: k that is not a multiple of 9 (the reference rejects it, src/main.c:61-63) ----
@pytest.mark.parametrize("k", [10, 13, 17, 22, 31, 32, 35, 40, 64, 125])
def test_host_index_any_k_ground_truth(hostlib, k):
    km = np.ascontiguousarray(S.distinct(S.kmers_of(S.random_genome(60000, 7 + k), k)))
    h = hostlib.bft_hosttest_build(km.ctypes.data, len(km), k, 0, 0)
    assert h
    rng = np.random.default_rng(1)
    q = np.concatenate([km, S.pack_codes(rng.integers(0, 4, (5000, k), dtype=np.uint8)), S.snp_mutants(km, k, 3)])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    bits = np.zeros((len(q) + 7) // 8, np.uint8)
    hostlib.bft_hosttest_query(h, q.ctypes.data, len(q), bits.ctypes.data, None)
    assert (S.from_bits(bits, len(q)) == S.member(q, km)).all()
    hostlib.bft_hosttest_free(h)


@pytest.mark.parametrize("k,levels", [(31, 1), (31, 3), (22, 2), (40, 3), (17, 1), (32, 1), (32, 2), (64, 2)])
def test_host_index_any_k_deep(hostlib, k, levels):
    km = np.ascontiguousarray(S.low_entropy_kmers(50000, k, 24, seed=k + levels, levels=levels))
    h = hostlib.bft_hosttest_build(km.ctypes.data, len(km), k, 0, 0)
    q = np.ascontiguousarray(np.concatenate([km, S.snp_mutants(km, k, 3)]))
    bits = np.zeros((len(q) + 7) // 8, np.uint8)
    hostlib.bft_hosttest_query(h, q.ctypes.data, len(q), bits.ctypes.data, None)
    assert (S.from_bits(bits, len(q)) == S.member(q, km)).all()
    hostlib.bft_hosttest_free(h)


def test_host_index_dense_remainder_groups(hostlib):
    """all 4^4 remainders under a few 27-mers (k=31) and tens of thousands of 8-nt remainders under one 9-mer (k=17)."""
    rng = np.random.default_rng(0)
    base = rng.integers(0, 4, (20, 27), dtype=np.uint8)
    rem = np.array([[(i >> 6) & 3, (i >> 4) & 3, (i >> 2) & 3, i & 3] for i in range(256)], dtype=np.uint8)
    c31 = np.concatenate([np.concatenate([np.repeat(base[j:j + 1], 256, 0), rem], axis=1) for j in range(20)])
    b9 = rng.integers(0, 4, (2, 9), dtype=np.uint8)
    c17 = np.concatenate([np.concatenate([np.repeat(b9[j:j + 1], 60000, 0), rng.integers(0, 4, (60000, 8), dtype=np.uint8)], axis=1) for j in range(2)])
    for k, codes in ((31, c31), (17, c17)):
        km = np.ascontiguousarray(S.distinct(S.pack_codes(codes)))
        h = hostlib.bft_hosttest_build(km.ctypes.data, len(km), k, 0, 0)
        q = np.ascontiguousarray(np.concatenate([km, S.snp_mutants(km, k, 3)]))
        bits = np.zeros((len(q) + 7) // 8, np.uint8)
        hostlib.bft_hosttest_query(h, q.ctypes.data, len(q), bits.ctypes.data, None)
        assert (S.from_bits(bits, len(q)) == S.member(q, km)).all()
        hostlib.bft_hosttest_free(h)


def test_group_shard_rule_through_the_c_symbol(built):
    """bft_gpu_group_shard (no GPU involved): slices are contiguous, cover [0, n), start on multiples of 64 k-mers (whole bytes of the
    presence bitmap) and are the ones dist.shard_bounds gives the ranks of a torch.distributed group."""
    from bloomfiltertrie_amd import bft as B
    from bloomfiltertrie_amd.dist import shard_bounds
    for n in (0, 1, 63, 64, 65, 1000, 12345, 10 ** 9, 10 ** 9 + 7):
        for parts in (1, 2, 3, 8):
            cover = 0
            for i in range(parts):
                a, b = B.shard(n, parts, i)
                assert a == cover and a <= b <= n and (a % 64 == 0 or a == n)
                assert (a, b) == shard_bounds(n, parts, i)[:2]
                cover = b
            assert cover == n
    with pytest.raises(Exception):
        B.shard(10, 0, 0)
    with pytest.raises(Exception):
        B.shard(10, 2, 2)


def test_hot_kernels_use_no_scratch(built):
    """Register budget of the kernels the headline runs on, read from the code objects of the built library (no GPU needed): the k-mer hash
    kernels must not touch scratch memory or spill vector registers.  (Round 3 lost 10 % of the headline for a while to a local whose
    address was taken: 4 bytes of scratch written per query, 534 MB per launch -- tools/kernel_resources.py shows it at once.)"""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_resources.py"), "_kh"], capture_output=True, text=True).stdout
    rows = [l.split(None, 7) for l in out.splitlines()[1:] if l.strip()]
    seen = set()
    for vgpr, sgpr, vspill, sspill, scratch, lds, maxwg, name in rows:
        m = re.match(r"void (k_query_kh|k_seq_kh|k_branching_kh|k_kh_assemble)<(\d)", name)
        if not m:
            continue
        seen.add(m.group(1))
        assert int(vspill) == 0 and int(scratch) == 0, name
        if m.group(1) == "k_query_kh":
            assert int(vgpr) <= 128, name   # (a lane's line passes through sixteen registers on its way from the quad: ~95; four wavefronts per SIMD at least)
    assert seen == {"k_query_kh", "k_seq_kh", "k_branching_kh", "k_kh_assemble"}


def test_build_and_resident_query_kernels_use_no_scratch(built):
    """A kernel that uses scratch memory starts ~0.13 ms late whenever the kernels before it on the queue used none (the runtime hands the queue's
    scratch back and sets it up again at the dispatch: DESIGN.md section 3, round 6).  The kernels one build and the resident colour queries launch --
    the sort's histogram and passes with atomic ranks (ranged, and chained with eight look-back groups: the forms the launcher picks), the scan, the
    bucket sorts of one- and two-word keys, the emits, the one-launch colour kernels -- must stay without; the ballot-rank fallbacks are launches
    of their own and may spill."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = set()
    for pat in ("k_rs_", "k_scan", "k_bucket", "k_colors_kh", "k_color_rows_kh", "k_msd", "k_fill_many", "k_prefix_scatter"):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_resources.py"), pat], capture_output=True, text=True).stdout
        for l in out.splitlines()[1:]:
            if not l.strip():
                continue
            vgpr, sgpr, vspill, sspill, scratch, lds, maxwg, name = l.split(None, 7)
            m = re.search(r"(k_rs_hist|k_rs_pass|k_rs_tiny|k_scan|k_bucket_sort_wave|k_bucket_sort|k_bucket2_sort_wave|k_bucket2_sort|k_bucket2_tiny|k_bucket_emit|k_bucket2_emit|"
                          r"k_colors_kh|k_color_rows_kh|k_msd_bounds|k_fill_many|k_prefix_scatter)<?([^(]*)", name)
            if not m:
                continue
            kern, targs = m.group(1), m.group(2)
            if kern == "k_rs_pass":  # <K, V, In, THREADS, IPT, RANGED, BALLOT, LBG>
                a = [x.strip() for x in targs.rstrip(">").split(",")]
                ranged, ballot, lbg = a[-3] == "true", a[-2] == "true", int(a[-1])
                if ballot or (not ranged and lbg == 1):
                    continue  # (fallback ranks; the one-group form is launched only where it spills nothing: bft_rs::uses_scratch)
            if kern == "k_rs_tiny" and targs.rstrip(">").split(",")[-1].strip() == "true":
                continue  # (ballot ranks)
            if kern in ("k_bucket_sort", "k_bucket2_sort") and targs.rstrip(">").split(",")[-1].strip() == "true":
                continue  # (the REDO launch: ballot ranks)
            if kern == "k_bucket2_sort_wave" and targs.rstrip(">").split(",")[-1].strip() == "false" and targs.startswith("32"):
                continue  # (ids of 2^18 and up at 32 items per lane: one wavefront per SIMD, AGPRs)
            seen.add(kern)
            assert int(scratch) == 0, (name[:160], scratch)
    assert {"k_rs_hist", "k_rs_pass", "k_scan", "k_bucket_sort_wave", "k_bucket_sort", "k_bucket2_sort_wave", "k_bucket_emit", "k_colors_kh", "k_color_rows_kh"} <= seen, seen
