"""Pin the oracle against every known answer available for the path:
 - the reference's own xxhash.c / popcnt.c / log2.c compiled into oracle/_ref (when present),
 - the published XXH64 test vector,
 - README.md:172 codec vector,
 - reference-run observations recorded in SURVEY.md / BASELINE.md (tests/golden/survey_pins.json).
"""
import json
import os
import random

import numpy as np
import pytest

from bloomfiltertrie_amd import synth as S

PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_pins.json")))


def test_xxh64_published_vectors(oracle_mod):
    O = oracle_mod
    assert O.xxh64(b"", 0) == 0xEF46DB3751D8E999  # xxHash spec, empty input, seed 0
    PRIME = 2654435761
    # the reference's own known-answer test: src/xxhsum.c:408-436 (BMK_sanityCheck; 32-bit generator)
    gen = PRIME
    buf = bytearray()
    for _ in range(101):
        buf.append((gen >> 24) & 0xFF)
        gen = (gen * gen) & 0xFFFFFFFF
    assert O.xxh64(b"", PRIME) == 0xAC75FDA2929B17EF
    assert O.xxh64(bytes(buf[:1]), 0) == 0x4FCE394CC88952D8
    assert O.xxh64(bytes(buf[:1]), PRIME) == 0x739840CB819FA723
    assert O.xxh64(bytes(buf[:14]), 0) == 0xCFFA8DB881BC3A3D
    assert O.xxh64(bytes(buf[:14]), PRIME) == 0x5B9611585EFCC9CB
    assert O.xxh64(bytes(buf[:101]), 0) == 0x0EAB543384F878AD
    assert O.xxh64(bytes(buf[:101]), PRIME) == 0xCAA65939306F1E21


def test_primitives_against_reference_sources(oracle_mod):
    O = oracle_mod
    ref = O.ref_prims()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(7)
    for seed in (0, 1, PINS["default_seeds"]["r1"], PINS["default_seeds"]["r2"], 2 ** 63 + 5):
        for ln in (0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 31, 32, 33, 63, 64, 65, 100, 257):
            d = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
            assert O.xxh64(d, seed) == ref.BFT_HASH_XXH64(d, ln, seed)
    for i in list(range(0, 70000, 7)) + [63, 64, 4095, 4096, 2 ** 18 - 1, 2 ** 18, 2 ** 24, 10 ** 8]:
        assert O.nb_bytes_id(i) == ref.get_nb_bytes_power2_annot(i), i
    import ctypes as C
    rev = (C.c_uint8 * 256).in_dll(ref, "rev")
    pop = (C.c_uint8 * 256).in_dll(ref, "POPCOUNT_8bit")
    for b in range(256):
        assert rev[b] == ((b & 3) << 6 | (b & 0xC) << 2 | (b & 0x30) >> 2 | (b & 0xC0) >> 6)
        assert pop[b] == bin(b).count("1")


def test_hash_v_survey_values(oracle_mod):
    t = oracle_mod.OracleBFT(27)
    hv = t.hash_v(4)
    assert hv[0] == PINS["hash_v"]["0"]
    assert hv[1] == PINS["hash_v"]["1"]
    assert hv[2] % 1504 == PINS["hash_v"]["2_mod_1504"]


def test_codec_readme_vector(oracle_mod):
    ok, p = oracle_mod.parse_kmer(PINS["codec"]["ascii"], 9)
    assert ok and [format(x, "08b") for x in p] == PINS["codec"]["bytes_bin"]
    assert oracle_mod.kmer_to_ascii(p, 9) == PINS["codec"]["ascii"]
    pk, valid = S.ascii_to_packed([PINS["codec"]["ascii"], "ACGTNACGT"], 9)
    assert valid.tolist() == [True, False] and (pk[0] == p).all() and not pk[1].any()
    ok2, p2 = oracle_mod.parse_kmer("ACGTNACGT", 9)
    assert not ok2


def test_config1_trie_shape_matches_reference_run(oracle_mod):
    """The reference run recorded in SURVEY.md section 6: 23 root CCs with nb_elem 24151, 23040, 20876 ...
    3417, 2610, 1191, 20 of them in p=14/s=4 mode, node UC of 166 rows, no child Node, 8.0 CCs scanned per hit."""
    pin = PINS["config1_trie"]
    random.seed(1)
    g = "".join(random.choice("ACGT") for _ in range(1000000))
    codes = S._CODE[np.frombuffer(g.encode(), dtype=np.uint8)]
    km = S.distinct(S.kmers_of(codes, pin["k"]))
    assert len(km) == pin["distinct_kmers"]
    t = oracle_mod.OracleBFT(pin["k"], count=True)
    t.insert_kmers(km, 0)
    st = t.stats()
    sizes = t.root_cc_sizes()
    assert st["root_ccs"] == pin["root_ccs"]
    assert sizes[:3] == pin["root_cc_nb_elem_first3"]
    assert sizes[-3:] == pin["root_cc_nb_elem_last3"]
    assert st["ccs_s4"] == pin["ccs_in_p14_s4_mode"]
    assert st["root_ccs"] - st["ccs_s4"] == pin["ccs_in_p10_s8_mode"]
    assert st["root_uc_rows"] == pin["root_uc_rows"]
    assert st["child_nodes"] == pin["child_nodes"]
    assert st["root_ccs"] * 188 == pin["bloom_filter_bytes_total"]
    assert st["kmers"] == pin["distinct_kmers"]
    rng = np.random.default_rng(0)
    q = km[rng.choice(len(km), 200000, replace=False)]
    bits, cnt = t.query_presence_count(q)
    assert S.from_bits(bits, len(q)).all()
    assert abs(cnt["ccs_scanned"] / len(q) - pin["mean_ccs_scanned_per_present_query"]) < 0.1


def test_annotation_mode_follows_the_insertion_history(oracle_mod):
    """a15: compute_best_mode (src/annotation.c:416-656) is applied at every insertion and keeps the current mode on a size tie
    (:652-653), so an annotation's bytes depend on the order its ids arrived in -- ascending, i.e. the sorted list replayed.
    The expected bytes below are derived by hand from the reference's rule (sizes of :621-633, choice of :638-653), NOT from
    either encoder:
      {6}      6 alone: mode 2 costs 1 byte, the bitmap CEIL(3+6, 8) = 2            -> mode 2: (6<<2)|2 = 0x1a
      {6,7}    at 7: list 2, ranges 1+1 = 2, bitmap CEIL(10/8) = 2: three-way tie, current mode 2 stays -> 0x1a 0x1e
               (a decision from scratch takes mode 0 on that tie: the bytes would be 00 03)
      {5,6}    5 alone: bitmap CEIL(8/8) = 1 <= list 1 -> mode 0; at 6: all three cost 2, mode 0 stays -> bits 7 and 8: 0x80 0x01
      {6,7,8}  at 8: list 3, ranges 2, bitmap 2: the current mode (2) is not minimal, mode 0 wins the 1-vs-0 tie -> bits 8,9,10: 00 07
      {6,8}    at 8: list 2, ranges 4, bitmap 2: tie between 2 and 0, current mode 2 stays -> 0x1a 0x22
      {70..73} two-byte ids: list 8 vs ranges 2+2 = 4 vs bitmap 10 -> mode 1: start 70 = 0x05 0x1a, end 73 = 0x05 0x26
    Both encoders (the oracle's and the product's, through libbft_hosttest.so) must give exactly these bytes, and every
    encoding must decode back to its id list."""
    import ctypes as C
    import os
    from bloomfiltertrie_amd import _lib
    lib = C.CDLL(os.path.join(_lib.CSRC, "libbft_hosttest.so"))
    lib.bft_hosttest_annot_encode.restype = C.c_int
    lib.bft_hosttest_annot_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    vectors = {(6,): "1a", (6, 7): "1a1e", (5, 6): "8001", (6, 7, 8): "0007", (6, 8): "1a22", (70, 71, 72, 73): "051a0526", (0,): "04", (5,): "80"}
    for ids, hexbytes in vectors.items():
        assert oracle_mod.annot_encode(list(ids)).hex() == hexbytes, ids
        a = np.array(ids, dtype=np.uint32)
        buf = np.zeros(64, np.uint8)
        n = lib.bft_hosttest_annot_encode(a.ctypes.data, len(a), buf.ctypes.data, 64)
        assert buf[:n].tobytes().hex() == hexbytes, ids
    rng = np.random.default_rng(11)
    for trial in range(400):  # round trips + agreement of the two encoders on random sets (dense, sparse, runs, ids up to 5000)
        kind = trial % 4
        if kind == 0:
            ids = np.flatnonzero(rng.random(rng.integers(1, 200)) < 0.6)
        elif kind == 1:
            ids = np.unique(rng.integers(0, 5000, rng.integers(1, 12)))
        elif kind == 2:
            a0 = int(rng.integers(0, 4200))
            ids = np.arange(a0, a0 + int(rng.integers(1, 90)))
        else:
            ids = np.unique(np.concatenate([np.arange(3, 3 + rng.integers(1, 9)), rng.integers(60, 300, 3)]))
        ids = ids.astype(np.uint32)
        if len(ids) == 0:
            continue
        enc = oracle_mod.annot_encode(ids.tolist())
        assert oracle_mod.annot_decode(enc) == ids.tolist()
        buf = np.zeros(16 + 4 * len(ids) + int(ids.max()) // 8, np.uint8)
        n = lib.bft_hosttest_annot_encode(ids.ctypes.data, len(ids), buf.ctypes.data, len(buf))
        assert n == len(enc) and buf[:n].tobytes() == enc
