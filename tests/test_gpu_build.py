"""GPU bulk build (k_pack_to_tform, radix sort, k_flags/k_scatter, bft_assemble.hip kernels) against the host
restatement of the same construction (csrc/bft_index.cpp, test helper library): every array of the image must be
bit-identical, on shallow, deep, tiny and multi-word-k inputs."""
import ctypes as C
import os

import numpy as np
import pytest

from bloomfiltertrie_amd import _lib, synth as S

pytestmark = pytest.mark.gpu

ARRAYS = ["tk", "nodes", "bfT", "ccs", "f2w", "clus", "child", "uck", "ucrow"]


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
