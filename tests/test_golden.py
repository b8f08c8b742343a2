"""Frozen golden vectors (tests/golden/bft_golden_k*.npz, made by tests/golden/make_golden.py from the definition of
the index with plain Python sets): the oracle on CPU, the HIP path on the GPU box."""
import glob
import os

import numpy as np
import pytest

from bloomfiltertrie_amd import synth as S

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "bft_golden_k*.npz")))


def _load(path):
    z = np.load(path)
    k, ngen = int(z["k"]), int(z["ngen"])
    return k, [z[f"genome_{g}"] for g in range(ngen)], z


def _check(t, z, q):
    bits, off, ids = t.query_colors(q)
    assert (bits == z["present_bits"]).all()
    assert (off == z["offsets"]).all() and (ids == z["ids"]).all()
    assert (t.query_presence(q) == z["present_bits"]).all()


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_against_golden(oracle_mod, path):
    k, genomes, z = _load(path)
    t = oracle_mod.OracleBFT(k)
    for g, km in enumerate(genomes):
        t.insert_kmers(km, g)
    assert t.stats()["child_nodes"] > 0 or k == 9
    q = z["queries"]
    _check(t, z, q)
    _, counts, nbr = t.query_branching(q)
    assert (counts == z["branching_counts"]).all()
    c = z["branching_counts"]
    assert nbr == int((((c >> 4) > 1) | ((c & 15) > 1)).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_gpu_against_golden(path):
    from bloomfiltertrie_amd import BFT
    k, genomes, z = _load(path)
    t = BFT(k)
    for g, km in enumerate(genomes):
        t.insert_kmers(km, g)
    q = z["queries"]
    _check(t, z, q)
    bits, counts = t.query_branching(q, with_counts=True)
    assert (counts == z["branching_counts"]).all()
    c = z["branching_counts"]
    assert (S.from_bits(bits, len(q)) == (((c >> 4) > 1) | ((c & 15) > 1))).all()
    _, rows = t.query_color_rows(q)
    unp = np.unpackbits(rows, axis=1, bitorder="little")[:, : len(genomes)]
    off, ids = z["offsets"], z["ids"]
    for i in range(0, len(q), 5):
        assert np.flatnonzero(unp[i]).tolist() == ids[int(off[i]):int(off[i + 1])].tolist()


def test_golden_files_exist():
    assert len(FILES) == 5
