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


def _ref_mode_history(ids):
    """compute_best_mode (src/annotation.c:416-656) replayed over ascending ids, written straight from the reference: while the
    annotation is a bitmap the two list sizes are re-derived by SCANNING ITS BITS (:476-535, with pow2_imin / tmp / tmp2 exactly
    as there), otherwise from the stored ids (:540-615).  Returns (mode, size) after the last insertion.  Independent of both encoders."""
    def nb(v):
        return (max(1, int(v).bit_length()) + 5) // 6

    def round_up(v):
        v -= 1
        for sh in (1, 2, 4, 8, 16):
            v |= v >> sh
        return v + 1

    def nb_bis(pos, pow2):
        return -(-(pow2.bit_length() - 1 + (1 if pow2 == pos else 0)) // 6)

    mode, size, have = None, 0, []
    for v in ids:
        tot1 = tot2 = 0
        tmp = tmp2 = 1
        last = None
        if mode == 0:
            bits = set(x + 2 for x in have)
            nbits = size * 8
            lim = nbits if nbits - 2 < 64 else 66
            run = False
            for i in range(2, lim):
                if i in bits:
                    last = i - 2
                    tot2 += 1
                    tot1 += 0 if run else 1
                    run = True
                else:
                    tot1 += 1 if run else 0
                    run = False
            if nbits - 2 < 64:
                tot1 += 1 if run else 0
            else:
                pow2 = 0
                for i in range(66, nbits):
                    if i in bits:
                        last = i - 2
                        if last >= pow2:
                            pow2 = round_up(last)
                            tmp = tmp2 = nb_bis(last, pow2)
                        tot2 += tmp
                        tot1 += 0 if run else tmp
                        run = True
                    else:
                        if run:
                            if i - 2 >= pow2:
                                pow2 = round_up(i - 2)
                                tmp2 = nb_bis(i - 2, pow2)
                            tot1 += tmp2
                        run = False
                if run:
                    tot1 += 1 if last < 0x40 else tmp
        elif mode in (1, 2):
            sizes = [nb(x) for x in have]
            last = have[-1]
            tot2 = sum(sizes)
            tot1 = sizes[0]
            for i in range(1, len(have)):
                if have[i] != have[i - 1] + 1:
                    tot1 += sizes[i - 1] + sizes[i]
            tot1 += sizes[-1]
        new0, new1, new2 = -(-(3 + v) // 8), tot1, tot2
        if last is None or v != last:
            if last is None or v != last + 1:
                new1 += 2 * nb(v)
            elif mode == 0:
                new1 += nb(v) - tmp
            else:
                new1 += nb(v) - nb(have[-1])
            new2 += nb(v)
        if new2 <= new1:
            m, sz = 2, new2
        else:
            m, sz = 1, new1
        if sz >= new0:
            m, sz = 0, new0
        if mode is not None and (new0, new1, new2)[mode] == sz and m != mode:
            m = mode
        mode, size = m, sz
        have.append(v)
    return mode, size


def test_annotation_run_end_estimate_in_bitmap_mode(oracle_mod):
    """a15, the run-end estimate (src/annotation.c:515-523): while an annotation is a bitmap, the reference prices the end of a run
    with the byte count of the id one past it -- one byte too many for a run that ends at 63 (or 4095, 262143).  Derived by hand:
      ids = 0..9, 20..29, 40..45, 50..55, 60..63: five runs of one-byte ids cost ranges 10 bytes, the list 36, the bitmap CEIL(66/8) = 9:
      a bitmap.  Inserting 110 (a new run of a two-byte id: + 4): ranges cost 14 exactly, the bitmap CEIL(113/8) = 15 -- an exact
      comparison leaves the bitmap (14 < 15); the reference adds 1 for the run that ends at 63 (priced as id 64: 2 bytes), sees 15 >= 15
      and KEEPS the bitmap: 15 bytes, bits id + 2.
    The same set with the last run ending at 62 instead has no such run: ranges 14 < 15 win.  Both encoders must agree with these and,
    on random sets around the byte-count edges 63/64 and 4095/4096, with the replay of the reference's own scan (_ref_mode_history)."""
    import ctypes as C
    import os
    from bloomfiltertrie_amd import _lib
    lib = C.CDLL(os.path.join(_lib.CSRC, "libbft_hosttest.so"))
    lib.bft_hosttest_annot_encode.restype = C.c_int
    lib.bft_hosttest_annot_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]

    def product(ids):
        a = np.array(ids, dtype=np.uint32)
        buf = np.zeros(64 + 4 * len(a) + int(a.max()) // 8, np.uint8)
        n = lib.bft_hosttest_annot_encode(a.ctypes.data, len(a), buf.ctypes.data, len(buf))
        return buf[:n].tobytes()

    runs63 = list(range(0, 10)) + list(range(20, 30)) + list(range(40, 46)) + list(range(50, 56)) + list(range(60, 64))
    ids = runs63 + [110]
    bitmap = bytearray(15)
    for g in ids:
        bitmap[(g + 2) // 8] |= 1 << ((g + 2) % 8)
    assert _ref_mode_history(ids) == (0, 15)
    assert oracle_mod.annot_encode(ids) == bytes(bitmap) and product(ids) == bytes(bitmap)
    ids62 = [g for g in runs63 if g != 63] + [110]   # the last run ends at 62: nothing is over-priced, ranges (14 bytes) win
    assert _ref_mode_history(ids62) == (1, 14)
    enc = oracle_mod.annot_encode(ids62)
    assert len(enc) == 14 and enc[0] & 3 == 1 and product(ids62) == enc and oracle_mod.annot_decode(enc) == ids62
    rng = np.random.default_rng(7)
    for trial in range(300):
        edge = (63, 4095)[trial % 2]
        parts = []
        for _ in range(int(rng.integers(2, 8))):   # a few runs below the edge, often one that ends on it or crosses it
            a0 = int(rng.integers(0, edge - 20))
            parts.append(np.arange(a0, a0 + int(rng.integers(1, 12))))
        end = edge + int(rng.integers(-1, 3))
        parts.append(np.arange(end - int(rng.integers(0, 6)), end + 1))
        parts.append(rng.integers(edge + 2, edge + 200, int(rng.integers(0, 4))))
        idl = np.unique(np.concatenate(parts)).astype(np.uint32).tolist()
        mode, size = _ref_mode_history(idl)
        enc = oracle_mod.annot_encode(idl)
        assert (enc[0] & 3, len(enc)) == (mode, size), (idl, mode, size)
        assert product(idl) == enc and oracle_mod.annot_decode(enc) == idl
