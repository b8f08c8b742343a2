"""ctypes wrapper of the CPU oracle (oracle/bft_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under bloomfiltertrie_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile liborc.so / liborc_count.so (and oracle/_ref when the reference is mounted)."""
    need = force or not all(os.path.exists(os.path.join(_HERE, f)) for f in ("liborc.so", "liborc_count.so"))
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("bft_oracle.c", "bft_oracle.h"))
    if not need:
        need = any(os.path.getmtime(os.path.join(_HERE, f)) < src_m for f in ("liborc.so", "liborc_count.so"))
    if need or (os.path.isdir("/root/reference/src") and not os.path.exists(os.path.join(_HERE, "_ref", "libbftref_prims.so"))):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def _load(count=False):
    name = "liborc_count.so" if count else "liborc.so"
    if name in _LIBS:
        return _LIBS[name]
    build()
    lib = C.CDLL(os.path.join(_HERE, name))
    u8p, u32p, u64p, lp, ip = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                               C.POINTER(C.c_long), C.POINTER(C.c_int))
    lib.orc_xxh64.restype = C.c_uint64
    lib.orc_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.orc_free.argtypes = [C.c_void_p]
    lib.orc_kmer_bytes.argtypes = [C.c_void_p]
    lib.orc_hash_v.restype = u64p
    lib.orc_hash_v.argtypes = [C.c_void_p]
    lib.orc_insert_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_uint32]
    lib.orc_freeze.argtypes = [C.c_void_p]
    lib.orc_query_presence.restype = C.c_long
    lib.orc_query_presence.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    lib.orc_query_presence_mt.restype = C.c_long
    lib.orc_query_presence_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int]
    lib.orc_query_presence_count.restype = C.c_long
    lib.orc_query_presence_count.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, u64p]
    lib.orc_query_colors.restype = C.c_long
    lib.orc_query_colors.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    lib.orc_parse_kmer.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    lib.orc_kmer_to_ascii.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    lib.orc_nb_bytes_id.argtypes = [C.c_uint32]
    lib.orc_annot_encode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.orc_annot_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.orc_stats.argtypes = [C.c_void_p, lp]
    lib.orc_root_cc_sizes.argtypes = [C.c_void_p, ip, C.c_int]
    lib.orc_extract.restype = C.c_long
    lib.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_colorset.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
    lib.orc_query_branching.restype = C.c_long
    lib.orc_query_branching.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    lib.orc_query_sequence.argtypes = [C.c_void_p, C.c_char_p, C.c_double, C.c_int, C.c_uint32, C.c_void_p, C.c_int]
    lib.orc_set_annotation_modes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.orc_write_bft.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.orc_load_bft.restype = C.c_void_p
    lib.orc_load_bft.argtypes = [C.c_char_p]
    lib.orc_nb_genomes_loaded.argtypes = [C.c_void_p]
    lib.orc_k.argtypes = [C.c_void_p]
    _LIBS[name] = lib
    return lib


def ref_prims():
    """oracle/_ref/libbftref_prims.so: the reference's own xxhash.c/popcnt.c/log2.c (or None)."""
    p = os.path.join(_HERE, "_ref", "libbftref_prims.so")
    if not os.path.exists(p):
        return None
    lib = C.CDLL(p)
    lib.BFT_HASH_XXH64.restype = C.c_uint64
    lib.BFT_HASH_XXH64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    lib.popcnt_8_par.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.get_nb_bytes_power2_annot.argtypes = [C.c_uint32]
    return lib


def xxh64(data: bytes, seed: int) -> int:
    return _load().orc_xxh64(data, len(data), seed)


def nb_bytes_id(i: int) -> int:
    return _load().orc_nb_bytes_id(i)


def annot_encode(ids) -> bytes:
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    out = np.zeros(8 + 4 * len(ids) + int(ids.max(initial=0)) // 8, dtype=np.uint8)
    n = _load().orc_annot_encode(ids.ctypes.data, len(ids), out.ctypes.data, len(out))
    assert n >= 0
    return out[:n].tobytes()


def annot_decode(b: bytes, cap=1 << 16):
    a = np.frombuffer(b, dtype=np.uint8).copy()
    out = np.zeros(cap, dtype=np.uint32)
    n = _load().orc_annot_decode(a.ctypes.data, len(a), out.ctypes.data, cap)
    return out[:n].tolist()


def parse_kmer(s: str, k: int):
    out = np.zeros((2 * k + 7) // 8, dtype=np.uint8)
    ok = _load().orc_parse_kmer(s.encode(), k, out.ctypes.data)
    return bool(ok), out


def kmer_to_ascii(kmer, k: int) -> str:
    kmer = np.ascontiguousarray(kmer, dtype=np.uint8)
    buf = C.create_string_buffer(k + 1)
    _load().orc_kmer_to_ascii(kmer.ctypes.data, k, buf)
    return buf.value.decode()


class OracleBFT:
    """Mirror of the reference calls on the path: createBFT_Root / insertKmers / isKmerPresent /
    get_annotation + get_list_id_genomes."""

    def __init__(self, k, r1=0, r2=0, count=False, _handle=None):
        self.lib = _load(count)
        self.h = _handle if _handle else self.lib.orc_create(k, r1, r2)
        if not self.h:
            raise ValueError("k must be a multiple of 9 in [9, 126] (reference src/main.c:61-63)")
        self.k = k
        self.nb = self.lib.orc_kmer_bytes(self.h)

    @classmethod
    def load_bft(cls, path, count=False):
        """read_BFT_Root (src/write_to_disk.c:260-776)."""
        lib = _load(count)
        h = lib.orc_load_bft(path.encode())
        if not h:
            raise ValueError(f"cannot load {path}")
        return cls(lib.orc_k(h), count=count, _handle=h)

    def write_bft(self, path, nb_genomes):
        """write_BFT_Root (src/write_to_disk.c:21-258)."""
        if self.lib.orc_write_bft(self.h, path.encode(), nb_genomes) != 0:
            raise IOError(path)

    def set_annotation_modes(self, comp=False, ext=False):
        """Writer test modes: mode-3 indices into comp_set_colors and/or extended-annotation bytes."""
        self.lib.orc_set_annotation_modes(self.h, int(comp), int(ext))

    def nb_genomes_loaded(self):
        return self.lib.orc_nb_genomes_loaded(self.h)

    def close(self):
        if self.h:
            self.lib.orc_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
        assert kmers.ndim == 2 and kmers.shape[1] == self.nb, (kmers.shape, self.nb)
        return kmers

    def hash_v(self, n=32):
        p = self.lib.orc_hash_v(self.h)
        return [p[i] for i in range(n)]

    def insert_kmers(self, kmers, id_genome):
        kmers = self._chk(kmers)
        self.lib.orc_insert_kmers(self.h, kmers.ctypes.data, len(kmers), id_genome)

    def freeze(self):
        self.lib.orc_freeze(self.h)

    def query_presence(self, kmers, threads=1):
        kmers = self._chk(kmers)
        bits = np.zeros((len(kmers) + 7) // 8, dtype=np.uint8)
        if threads > 1:
            self.lib.orc_query_presence_mt(self.h, kmers.ctypes.data, len(kmers), bits.ctypes.data, threads)
        else:
            self.lib.orc_query_presence(self.h, kmers.ctypes.data, len(kmers), bits.ctypes.data)
        return bits

    def query_presence_count(self, kmers):
        kmers = self._chk(kmers)
        bits = np.zeros((len(kmers) + 7) // 8, dtype=np.uint8)
        out = (C.c_uint64 * 3)()
        self.lib.orc_query_presence_count(self.h, kmers.ctypes.data, len(kmers), bits.ctypes.data, out)
        return bits, {"bytes": int(out[0]), "ccs_scanned": int(out[1]), "levels": int(out[2])}

    def query_colors(self, kmers):
        kmers = self._chk(kmers)
        n = len(kmers)
        bits = np.zeros((n + 7) // 8, dtype=np.uint8)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        cap = max(1024, 4 * n)
        while True:
            ids = np.zeros(cap, dtype=np.uint32)
            tot = self.lib.orc_query_colors(self.h, kmers.ctypes.data, n, bits.ctypes.data, offsets.ctypes.data,
                                            ids.ctypes.data, cap)
            if tot <= cap:
                return bits, offsets, ids[:tot]
            cap = tot

    def query_branching(self, kmers):
        kmers = self._chk(kmers)
        bits = np.zeros((len(kmers) + 7) // 8, dtype=np.uint8)
        counts = np.zeros(len(kmers), dtype=np.uint8)
        n = self.lib.orc_query_branching(self.h, kmers.ctypes.data, len(kmers), bits.ctypes.data, counts.ctypes.data)
        return bits, counts, int(n)

    def query_sequence(self, seq, threshold, canonical, nb_genomes):
        out = np.zeros(max(1, nb_genomes), dtype=np.uint32)
        n = self.lib.orc_query_sequence(self.h, seq.encode(), float(threshold), int(canonical), nb_genomes, out.ctypes.data, len(out))
        return out[:n].tolist()

    def stats(self):
        out = (C.c_long * 10)()
        self.lib.orc_stats(self.h, out)
        names = ["nodes", "ccs", "kmers", "root_ccs", "root_uc_rows", "uc_rows", "child_nodes", "prefixes",
                 "ccs_s4", "max_ccs_per_node"]
        return dict(zip(names, [int(x) for x in out]))

    def root_cc_sizes(self):
        out = (C.c_int * 4096)()
        n = self.lib.orc_root_cc_sizes(self.h, out, 4096)
        return [int(out[i]) for i in range(n)]

    def extract(self):
        n = self.lib.orc_extract(self.h, None, None)
        kmers = np.zeros((n, self.nb), dtype=np.uint8)
        cs = np.zeros(n, dtype=np.uint32)
        self.lib.orc_extract(self.h, kmers.ctypes.data, cs.ctypes.data)
        return kmers, cs

    def colorset(self, cs):
        out = np.zeros(1 << 16, dtype=np.uint32)
        n = self.lib.orc_colorset(self.h, int(cs), out.ctypes.data, len(out))
        return out[:n].tolist()
