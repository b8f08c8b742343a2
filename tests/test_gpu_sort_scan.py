"""The library's own device-wide primitives (csrc/bft_sort.h, csrc/bft_scan.h) against torch on the GPU, through two test hooks of
libbft_gpu.so (bft_gpu_test_sort / bft_gpu_test_scan: not part of the C-ABI header).  The build uses them everywhere a library sort or scan
used to run (root-prefix split, k-mer hash sort, the assembly's and the interning's sorts, every offsets array); the builds' images are
compared elsewhere (tests/test_gpu_build.py) -- here the primitives themselves, on ragged sizes in each of the sort's three regimes (one tile
in LDS; a histogram + ranged pass per digit; ranged first pass + look-back passes in chains), every tile shape, partial bit ranges, and the
stability an LSD sort lives on."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from bloomfiltertrie_amd import _lib
    L = _lib.load()
    L.bft_gpu_test_sort.restype = C.c_int
    L.bft_gpu_test_sort.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bft_gpu_test_scan.restype = C.c_int
    L.bft_gpu_test_scan.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    return L


SIZES = [1, 2, 63, 64, 65, 4097, 12288, 12289, 100_003, 1_000_003, 9_000_001, 20_000_003]  # (2^23 = 8.4 x 10^6: the chained regime starts there)


@pytest.mark.parametrize("kind,shape,bits", [(0, 0, (0, 64)), (0, 0, (43, 61)), (1, 0, (0, 63)), (1, 1, (5, 33)), (1, 2, (0, 24)), (2, 0, (0, 32)), (2, 2, (3, 27)), (2, 1, (0, 9))])
def test_sort_is_the_stable_sort_of_the_bit_range(lib, kind, shape, bits):
    import torch
    dev = torch.device("cuda", 0)
    b0, b1 = bits
    g = torch.Generator(device=dev)
    g.manual_seed(kind * 100 + shape * 10 + b0)
    for n in SIZES:
        if kind == 2:
            keys = torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
            wide = keys.to(torch.int64)
        else:
            keys = torch.randint(0, 2**62, (n,), dtype=torch.int64, device=dev, generator=g)
            if n > 1000:
                keys[::7] = keys[3]  # long runs of equal keys: what stability is about
            wide = keys
        vals = torch.arange(n, dtype=torch.int32, device=dev)
        digit = (wide >> b0) & ((1 << (b1 - b0)) - 1)
        order = torch.sort(digit, stable=True).indices
        out_k = torch.empty_like(keys)
        out_v = torch.empty_like(vals)
        rc = lib.bft_gpu_test_sort(kind, shape, keys.data_ptr(), vals.data_ptr() if kind else None, n, b0, b1, out_k.data_ptr(), out_v.data_ptr() if kind else None, None)
        assert rc == 0, (n, rc)
        torch.cuda.synchronize()
        assert torch.equal(out_k, keys[order]), (kind, shape, bits, n)
        if kind:
            assert torch.equal(out_v, vals[order]), (kind, shape, bits, n)


def test_sort_with_ballot_ranks(lib):
    """"sort_ballots" 1: the ranks of a tile from wavefront ballots (what a device that does not serve an LDS atomic's lanes in lane order would run),
    in all three regimes -- the chained passes with one and with eight look-back groups included."""
    import torch
    from bloomfiltertrie_amd import BFT
    dev = torch.device("cuda", 0)
    w = BFT(27)
    try:
        w.set_option("sort_ballots", 1)
        g = torch.Generator(device=dev)
        g.manual_seed(99)
        for kind, shape, (b0, b1) in ((0, 0, (43, 61)), (1, 0, (0, 40)), (2, 2, (0, 27)), (1, 1, (7, 30))):
            for n in (100, 12289, 700_001, 9_000_001, 20_000_003):
                if kind == 2:
                    keys = torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
                    wide = keys.to(torch.int64)
                else:
                    keys = torch.randint(0, 2**62, (n,), dtype=torch.int64, device=dev, generator=g)
                    keys[::5] = keys[1]
                    wide = keys
                vals = torch.arange(n, dtype=torch.int32, device=dev)
                order = torch.sort((wide >> b0) & ((1 << (b1 - b0)) - 1), stable=True).indices
                out_k, out_v = torch.empty_like(keys), torch.empty_like(vals)
                assert lib.bft_gpu_test_sort(kind, shape, keys.data_ptr(), vals.data_ptr() if kind else None, n, b0, b1, out_k.data_ptr(), out_v.data_ptr() if kind else None, None) == 0
                torch.cuda.synchronize()
                assert torch.equal(out_k, keys[order]), (kind, shape, n)
                if kind:
                    assert torch.equal(out_v, vals[order]), (kind, shape, n)
    finally:
        w.set_option("sort_ballots", 0)
        w.close()


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_scans_and_their_totals(lib, kind):
    """Exclusive sums of u32 and u64 (with the total behind the last element and in a slot of its own) and the inclusive running maximum; three scans
    in a row share one scratch block -- a launch zeroes the states of the launch before it, no memset in between -- and a second stream of calls
    with other sizes follows on the same block sizes (each call of the hook has its own block: the sharing is inside a call)."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(kind)
    for n in [1, 63, 4096, 4097, 65_537, 1_000_003, 30_000_001]:
        if kind == 0:
            x = torch.randint(0, 7, (n,), dtype=torch.int32, device=dev, generator=g)
            out = torch.full((n + 1,), -1, dtype=torch.int32, device=dev)
            want = torch.cumsum(x.to(torch.int64), 0)
            exp = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), want]).to(torch.int32)
            total = int(want[-1])
        elif kind == 1:
            x = torch.randint(0, 2**33, (n,), dtype=torch.int64, device=dev, generator=g)
            out = torch.full((n + 1,), -1, dtype=torch.int64, device=dev)
            want = torch.cumsum(x, 0)
            exp = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), want])
            total = int(want[-1])
        else:
            x = torch.randint(0, 2**40, (n,), dtype=torch.int64, device=dev, generator=g)
            out = torch.full((n,), -1, dtype=torch.int64, device=dev)
            exp = torch.cummax(torch.clamp(x, min=5), 0).values
            total = int(exp[-1])
        tot = torch.zeros(1, dtype=torch.int64, device=dev)
        assert lib.bft_gpu_test_scan(kind, x.data_ptr(), n, out.data_ptr(), tot.data_ptr(), 3, None) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, exp), (kind, n)
        assert int(tot[0]) == total, (kind, n)
