"""The reference's .bft file format, CPU side:
 - oracle writer/reader round trip (byte-identical rewrite), incl. level_min==0 levels and child Nodes;
 - the product's reader (csrc/bft_file.cpp) on oracle-written files: k-mers and colour sets recovered exactly;
 - the product's writer on a host-built image, loaded back by the oracle and queried with the restated reference
   algorithm (proves the bulk-built containers satisfy the invariants of SURVEY.md A.7);
 - the size of the config-1 file against the reference run recorded in BASELINE.md (6.36 MB)."""
import ctypes as C
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from bloomfiltertrie_amd import _lib, synth as S


@pytest.fixture(scope="module")
def hostlib():
    subprocess.check_call(["make", "-C", _lib.CSRC, "libbft_hosttest.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(_lib.CSRC, "libbft_hosttest.so"))
    lib.bft_hosttest_build.restype = C.c_void_p
    lib.bft_hosttest_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int]
    lib.bft_hosttest_free.argtypes = [C.c_void_p]
    lib.bft_hosttest_write_bft.argtypes = [C.c_void_p, C.c_char_p]
    lib.bft_hosttest_read_bft.restype = C.c_void_p
    lib.bft_hosttest_read_bft.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    lib.bft_hosttest_read_genome.restype = C.c_uint64
    lib.bft_hosttest_read_genome.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.bft_hosttest_read_free.argtypes = [C.c_void_p]
    return lib


def _product_read(hostlib, path):
    k, g, n = C.c_int(), C.c_int(), C.c_uint64()
    h = hostlib.bft_hosttest_read_bft(path.encode(), C.byref(k), C.byref(g), C.byref(n))
    assert h, "product reader rejected the file"
    nb = (2 * k.value + 7) // 8
    out = []
    for gi in range(g.value):
        sz = hostlib.bft_hosttest_read_genome(h, gi, None)
        buf = np.zeros(sz, np.uint8)
        hostlib.bft_hosttest_read_genome(h, gi, buf.ctypes.data)
        out.append(buf.reshape(-1, nb))
    hostlib.bft_hosttest_read_free(h)
    return k.value, n.value, out


CASES = [(9, 1, 1), (18, 1, 3), (27, 2, 1), (27, 1, 6), (36, 3, 3), (45, 2, 2), (63, 3, 70), (126, 2, 2)]


@pytest.mark.parametrize("k,levels,ngen", CASES)
def test_oracle_roundtrip_and_product_reader(oracle_mod, hostlib, tmp_path, k, levels, ngen):
    base = S.low_entropy_kmers(30000, k, 16, seed=k + levels, levels=levels)
    gk = [base[:: (g % 5) + 1] for g in range(ngen)]
    a = oracle_mod.OracleBFT(k)
    for g, km in enumerate(gk):
        a.insert_kmers(km, g)
    p1, p2 = str(tmp_path / "a.bft"), str(tmp_path / "b.bft")
    a.write_bft(p1, ngen)
    b = oracle_mod.OracleBFT.load_bft(p1)
    assert b.nb_genomes_loaded() == ngen and a.stats() == b.stats()
    b.write_bft(p2, ngen)
    assert open(p1, "rb").read() == open(p2, "rb").read()
    q = np.concatenate([base, S.snp_mutants(base, k, 1)])
    ra, rb = a.query_colors(q), b.query_colors(q)
    assert all((x == y).all() for x, y in zip(ra, rb))
    # product reader: per-genome k-mer sets
    kk, n, per = _product_read(hostlib, p1)
    assert kk == k and n == len(base) and len(per) == ngen
    for g in range(ngen):
        assert sorted(S.row_keys(per[g]).tolist()) == sorted(S.row_keys(gk[g]).tolist())


@pytest.mark.parametrize("k,levels,n", [(9, 1, 20000), (18, 1, 30000), (27, 1, 200000), (27, 2, 40000), (36, 3, 40000), (63, 3, 40000), (45, 4, 40000), (27, 1, 100), (27, 1, 255)])
def test_product_writer_loaded_by_oracle(oracle_mod, hostlib, tmp_path, k, levels, n):
    km = S.low_entropy_kmers(n, k, 24, seed=n + k, levels=levels) if n > 1000 else S.distinct(S.pack_codes(np.random.default_rng(n).integers(0, 4, (n, k), dtype=np.uint8)))
    if k == 27 and n == 200000:
        km = S.distinct(S.kmers_of(S.random_genome(n, 3), k))  # a large root with both filter2 geometries
    km = np.ascontiguousarray(km)
    h = hostlib.bft_hosttest_build(km.ctypes.data, len(km), k, 0, 0)
    path = str(tmp_path / "p.bft")
    assert hostlib.bft_hosttest_write_bft(h, path.encode()) == 0
    hostlib.bft_hosttest_free(h)
    o = oracle_mod.OracleBFT.load_bft(path)  # the restated reference reader + query algorithm on OUR containers
    assert o.stats()["kmers"] == len(km)
    rng = np.random.default_rng(0)
    q = np.concatenate([km, S.snp_mutants(km, k, 2), S.pack_codes(rng.integers(0, 4, (2000, k), dtype=np.uint8))])
    bits, off, ids = o.query_colors(q)
    truth = S.member(q, km)
    assert (S.from_bits(bits, len(q)) == truth).all()
    assert (ids == 0).all() and len(ids) == truth.sum()
    # and the product reader reads its own file
    kk, nk, per = _product_read(hostlib, path)
    assert nk == len(km) and sorted(S.row_keys(per[0]).tolist()) == sorted(S.row_keys(km).tolist())


@pytest.mark.parametrize("k,levels,n", [(27, 1, 120000), (36, 2, 60000), (63, 3, 30000)])
def test_writer_and_reader_do_not_depend_on_the_thread_count(hostlib, tmp_path, k, levels, n):
    """The writer emits the file as ordered parts filled by a pool of host threads (UC blocks and child-Node subtrees of the root's CCs) and the
    loader rebuilds the k-mers block by block in parallel (csrc/bft_file.cpp; BFT_GPU_IO_THREADS): one thread, three and the default give the same
    bytes, and the same k-mers back."""
    km = np.ascontiguousarray(S.low_entropy_kmers(n, k, 40, seed=n + k, levels=levels))
    h = hostlib.bft_hosttest_build(km.ctypes.data, len(km), k, 0, 0)
    blobs, reads = [], []
    try:
        for nt in ("1", "3", None):
            if nt is None:
                os.environ.pop("BFT_GPU_IO_THREADS", None)
            else:
                os.environ["BFT_GPU_IO_THREADS"] = nt
            path = str(tmp_path / f"t{nt}.bft")
            assert hostlib.bft_hosttest_write_bft(h, path.encode()) == 0
            blobs.append(open(path, "rb").read())
            kk, nk, per = _product_read(hostlib, path)
            assert nk == len(km)
            reads.append(np.sort(S.row_keys(per[0])))
    finally:
        os.environ.pop("BFT_GPU_IO_THREADS", None)
        hostlib.bft_hosttest_free(h)
    assert blobs[0] == blobs[1] == blobs[2]
    assert (reads[0] == reads[1]).all() and (reads[0] == reads[2]).all() and (reads[0] == np.sort(S.row_keys(km))).all()


def test_config1_file_size_matches_reference_run(oracle_mod, tmp_path):
    """`bft build 27` on the config-1 genome wrote a 6.36 MB .bft."""
    random.seed(1)
    g = "".join(random.choice("ACGT") for _ in range(1000000))
    km = S.distinct(S.kmers_of(S._CODE[np.frombuffer(g.encode(), dtype=np.uint8)], 27))
    t = oracle_mod.OracleBFT(27)
    t.insert_kmers(km, 0)
    p = str(tmp_path / "c1.bft")
    t.write_bft(p, 1)
    assert 6.355e6 <= os.path.getsize(p) < 6.365e6


def test_reader_rejects_garbage(hostlib, tmp_path):
    p = str(tmp_path / "bad.bft")
    open(p, "wb").write(b"\x00" * 10)
    k, g, n = C.c_int(), C.c_int(), C.c_uint64()
    assert not hostlib.bft_hosttest_read_bft(p.encode(), C.byref(k), C.byref(g), C.byref(n))


@pytest.mark.parametrize("comp,ext", [(True, False), (False, True), (True, True)])
@pytest.mark.parametrize("k,levels,ngen", [(27, 1, 9), (36, 3, 70), (18, 1, 200)])
def test_reference_shaped_annotations(oracle_mod, hostlib, tmp_path, k, levels, ngen, comp, ext):
    """Files as the reference's CLI leaves them: mode-3 annotations indexing comp_set_colors (>= 7 genomes,
    src/file_io.c:192-193) and extended-annotation bytes (src/UC.c:321-521).  Read back by the oracle's reader and by
    the product's reader."""
    base = S.low_entropy_kmers(20000, k, 16, seed=k + levels, levels=levels)
    rng = np.random.default_rng(ngen)
    gk = [base[rng.random(len(base)) < rng.uniform(0.05, 0.9)] for _ in range(ngen)]
    a = oracle_mod.OracleBFT(k)
    for g, km in enumerate(gk):
        a.insert_kmers(np.ascontiguousarray(km), g)
    q = np.concatenate([base, S.snp_mutants(base[::3], k, 1)])
    exp = a.query_colors(q)
    a.set_annotation_modes(comp=comp, ext=ext)
    got = a.query_colors(q)  # the oracle itself now decodes through comp_set_colors
    assert all((x == y).all() for x, y in zip(got, exp))
    p = str(tmp_path / "r.bft")
    a.write_bft(p, ngen)
    plain = str(tmp_path / "plain.bft")
    a.set_annotation_modes(False, False)
    a.write_bft(plain, ngen)
    if comp:
        assert os.path.getsize(p) != os.path.getsize(plain)
    b = oracle_mod.OracleBFT.load_bft(p)
    assert all((x == y).all() for x, y in zip(b.query_colors(q), exp))
    kk, n, per = _product_read(hostlib, p)
    assert kk == k and n == len(S.distinct(np.concatenate(gk)))
    for g in range(ngen):
        assert sorted(S.row_keys(per[g]).tolist()) == sorted(S.row_keys(gk[g]).tolist())


def test_reader_survives_mutated_files(hostlib):
    """tools/fuzz_bft_reader.py: truncated files, flipped bytes and extreme values in 2- and 4-byte fields of valid files at five k / depth
    combinations -- the reader rejects or accepts each one without crashing (a batch runs in a process of its own: a crash fails the run)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_bft_reader.py"), "2", "120", "7"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    assert b"fuzz OK" in r.stdout


_STARVED_READER = r"""
import ctypes as C, os, resource, sys
lib = C.CDLL(sys.argv[1])
lib.bft_hosttest_read_bft.restype = C.c_void_p
lib.bft_hosttest_read_bft.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
lib.bft_hosttest_read_free.argtypes = [C.c_void_p]
def vm_bytes():
    return int(open("/proc/self/statm").read().split()[0]) * os.sysconf("SC_PAGE_SIZE")
def read():
    k, g, n = C.c_int(), C.c_int(), C.c_uint64()
    h = lib.bft_hosttest_read_bft(sys.argv[2].encode(), C.byref(k), C.byref(g), C.byref(n))
    if h:
        lib.bft_hosttest_read_free(h)
    return bool(h), n.value
soft0, hard = resource.getrlimit(resource.RLIMIT_AS)
res = []
resource.setrlimit(resource.RLIMIT_STACK, (512 << 10, resource.getrlimit(resource.RLIMIT_STACK)[1]))  # (small stacks for the reader's threads: what is short is the heap)
for extra_mb in (112, 104, 96, 64):  # address space left to the reader: it needs ~120 MB for this file
    resource.setrlimit(resource.RLIMIT_AS, (vm_bytes() + (extra_mb << 20), hard))
    try:
        ok, _ = read()
    except MemoryError:
        ok = False
    resource.setrlimit(resource.RLIMIT_AS, (soft0, hard))
    res.append(ok)
ok, n = read()  # the limit lifted: the library is as good as new
print("starved", res, "final", ok, n)
"""


def test_reader_out_of_memory_in_its_worker_threads_is_an_error_not_the_end_of_the_process(oracle_mod, hostlib, tmp_path):
    """A small file whose every k-mer is annotated with every genome (the reference writes a range annotation of a few bytes per row;
    the reader's threads expand it to nb_genomes x k-mers x B bytes of per-genome output).  With the address space capped just above what the
    process already holds, the allocations of the worker threads fail: the call returns an error -- no std::terminate from a thread, no joinable
    thread left behind -- and the same process reads the file once the cap is lifted."""
    k, ngen = 27, 96
    km = S.distinct(S.kmers_of(S.random_genome(120000, 4), k))
    a = oracle_mod.OracleBFT(k)
    for g in range(ngen):
        a.insert_kmers(km, g)
    p = str(tmp_path / "wide.bft")
    a.write_bft(p, ngen)
    assert os.path.getsize(p) < 0.02 * ngen * len(km) * 7  # (the file is tiny beside what it decodes to)
    r = subprocess.run([sys.executable, "-c", _STARVED_READER, os.path.join(_lib.CSRC, "libbft_hosttest.so"), p], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, BFT_GPU_IO_THREADS="3"))  # (few threads: glibc aborts a process whose NEW thread cannot get its thread-local block)
    out = r.stdout.decode(errors="replace")
    if r.returncode != 0 and "cannot allocate memory for thread-local data" in out:
        pytest.skip("glibc itself aborts a process whose new thread cannot get its thread-local block: the cap hit that allocation, not the reader's")
    assert r.returncode == 0, out[-2000:]
    line = [ln for ln in out.splitlines() if ln.startswith("starved")][-1]
    assert "False" in line.split("final")[0], line        # at least one capped read failed cleanly ...
    assert f"final True {len(km)}" in line, line           # ... and the process still reads the file afterwards
