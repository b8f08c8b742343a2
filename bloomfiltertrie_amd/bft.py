"""Host-side mirror of the reference interface for the batched path (include/bft.h, include/insertNode.h,
include/presenceNode.h of GuillaumeHolley/BloomFilterTrie), over the C-ABI of include/bft_gpu.h.

Names follow the reference: create_cdbg / insert_kmers_new_genome / insertKmers / is present / get_annotation +
get_list_id_genomes.  Batches are numpy uint8 arrays [n, CEIL(2k/8)] in the reference's packed layout
(bloomfiltertrie_amd.synth / src/fasta.c:3-53).
"""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import _lib
from .synth import ascii_to_packed, kmer_bytes

_LIVE = weakref.WeakSet()


@atexit.register
def _close_all():
    """Free every live handle before the interpreter (and the HIP runtime under it) is torn down."""
    for t in list(_LIVE):
        try:
            t.close()
        except Exception:
            pass


INFO_FIELDS = ["k", "kmers", "nodes", "ccs", "uc_rows", "child_nodes", "prefixes", "ccs_s4", "max_ccs_per_node",
               "pairs", "colorsets", "genomes", "image_bytes", "root_ccs", "root_uc_rows", "pending_pairs"]


class BFT:
    """One Bloom Filter Trie resident in the HBM of one MI355X (replaces BFT_Root, include/Node.h:96-122)."""

    def __init__(self, k, device=0, r1=0, r2=0, _handle=None):
        self._lib = _lib.load()
        h = C.c_void_p()
        if _handle is not None:
            h = _handle
        else:
            _lib.check(self._lib.bft_gpu_create_seeded(k, device, r1, r2, C.byref(h)))  # create_cdbg, include/bft.h:62
        self._h = h
        self.k = k
        self.nb = kmer_bytes(k)
        self.device = device
        _LIVE.add(self)

    @classmethod
    def load_bft(cls, path, device=0):
        """load_BFT (include/bft.h:176)."""
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.bft_gpu_load_bft(path.encode(), device, C.byref(h)))
        out = (C.c_uint64 * 16)()
        _lib.check(lib.bft_gpu_info(h, out, 16))
        return cls(int(out[0]), device=device, _handle=h)

    @classmethod
    def from_image(cls, d_blob_ptr, nbytes, device=0):
        """New index on `device` from an image blob resident in that GPU's memory (see image_pack)."""
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.bft_gpu_image_unpack(d_blob_ptr, nbytes, device, C.byref(h)))
        out = (C.c_uint64 * 16)()
        _lib.check(lib.bft_gpu_info(h, out, 16))
        return cls(int(out[0]), device=device, _handle=h)

    def image_size(self):
        """Bytes of the device blob that image_pack writes."""
        n = C.c_uint64()
        _lib.check(self._lib.bft_gpu_image_size(self._h, C.byref(n)))
        return int(n.value)

    def image_pack(self, d_blob_ptr, cap, stream=None):
        """Copy the built index (containers, k-mer table, colour sets, genome names) into one
        contiguous device buffer -- the payload of the broadcast that replicates the trie on the other GPUs."""
        _lib.check(self._lib.bft_gpu_image_pack(self._h, d_blob_ptr, cap, stream))

    def write_bft(self, path):
        """write_BFT (include/bft.h:175)."""
        _lib.check(self._lib.bft_gpu_write_bft(self._h, path.encode()))

    # -- lifecycle ----------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.bft_gpu_free(self._h)  # free_cdbg, include/bft.h:63
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
        if kmers.ndim != 2 or kmers.shape[1] != self.nb:
            raise ValueError(f"expected packed k-mers of shape [n, {self.nb}], got {kmers.shape}")
        return kmers

    # -- insertion ----------------------------------------------------------------------------------------------
    def add_genome(self, name):
        """add_genomes_BFT_Root (include/CC.h:307-338)."""
        gid = C.c_uint32()
        _lib.check(self._lib.bft_gpu_add_genome(self._h, name.encode(), C.byref(gid)))
        return gid.value

    def insert_kmers(self, kmers, id_genome):
        """insertKmers(root, array_kmers, nb_kmers, id_genome, size_id_genome) (include/insertNode.h:26)."""
        kmers = self._chk(kmers)
        _lib.check(self._lib.bft_gpu_insert_kmers(self._h, kmers.ctypes.data, len(kmers), id_genome))

    def insert_kmers_dev(self, d_ptr, n, id_genome):
        _lib.check(self._lib.bft_gpu_insert_kmers_dev(self._h, d_ptr, n, id_genome))

    def insert_kmers_dev_async(self, d_ptr, n, id_genome, stream):
        """insert_kmers_dev, stream-ordered on the caller's HIP stream (raw handle; None / 0 = the null stream): returns at once."""
        _lib.check(self._lib.bft_gpu_insert_kmers_dev_async(self._h, d_ptr, n, id_genome, stream))

    def insert_kmers_new_genome(self, kmers_ascii, genome_name):
        """insert_kmers_new_genome(nb_kmers, kmers, genome_name, bft) (include/bft.h:72)."""
        gid = self.add_genome(genome_name)
        packed, valid = ascii_to_packed(kmers_ascii, self.k)
        if not valid.all():
            raise ValueError("k-mer with a character outside ACGTU (reference get_kmer/insert path exits, src/bft.c:239)")
        self.insert_kmers(packed, gid)
        return gid

    def build(self):
        _lib.check(self._lib.bft_gpu_build(self._h))

    # -- queries ------------------------------------------------------------------------------------------------
    def query_presence(self, kmers):
        """Loop of src/file_io.c:726-768 over isKmerPresent: presence bitmap (bit i%8 of byte i//8)."""
        kmers = self._chk(kmers)
        bits = np.zeros((len(kmers) + 7) // 8, dtype=np.uint8)
        _lib.check(self._lib.bft_gpu_query_presence(self._h, kmers.ctypes.data, len(kmers), bits.ctypes.data))
        return bits

    def query_presence_dev(self, d_kmers_ptr, n, d_bits_ptr, stream=None):
        """Device-resident variant: pointers into HBM (e.g. torch tensor .data_ptr()); asynchronous on `stream`."""
        _lib.check(self._lib.bft_gpu_query_presence_dev(self._h, d_kmers_ptr, n, d_bits_ptr, stream))

    def query_colors(self, kmers):
        """get_annotation + get_list_id_genomes per k-mer: (bits, offsets[n+1], ids)."""
        kmers = self._chk(kmers)
        n = len(kmers)
        bits = np.zeros((n + 7) // 8, dtype=np.uint8)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        cap = max(1024, 4 * n)
        while True:
            ids = np.zeros(cap, dtype=np.uint32)
            need = C.c_uint64()
            rc = self._lib.bft_gpu_query_colors(self._h, kmers.ctypes.data, n, bits.ctypes.data, offsets.ctypes.data,
                                                ids.ctypes.data, cap, C.byref(need))
            if rc == -6:  # BFT_GPU_E_NOSPACE
                cap = int(need.value)
                continue
            _lib.check(rc)
            return bits, offsets, ids[:int(need.value)]

    def query_rows(self, kmers):
        """What resultPresence holds for each k-mer (include/Node.h:60-92) as indexes: (bits, row in the stored k-mer
        table, colour-set id); 0xFFFFFFFF where absent."""
        kmers = self._chk(kmers)
        n = len(kmers)
        bits = np.zeros((n + 7) // 8, dtype=np.uint8)
        rows = np.zeros(n, dtype=np.uint32)
        sets = np.zeros(n, dtype=np.uint32)
        _lib.check(self._lib.bft_gpu_query_rows(self._h, kmers.ctypes.data, n, bits.ctypes.data, rows.ctypes.data, sets.ctypes.data))
        return bits, rows, sets

    def query_color_rows(self, kmers):
        """Fixed-width colour rows (the CSV row of src/file_io.c:744-765 before formatting)."""
        kmers = self._chk(kmers)
        n = len(kmers)
        self.build()  # the genome count is known once the image exists
        g = self.info()["genomes"]
        bits = np.zeros((n + 7) // 8, dtype=np.uint8)
        rows = np.zeros((n, (g + 7) // 8), dtype=np.uint8)
        _lib.check(self._lib.bft_gpu_query_color_rows(self._h, kmers.ctypes.data, n, bits.ctypes.data, rows.ctypes.data))
        return bits, rows

    def query_colors_dev(self, d_kmers_ptr, n, d_bits_ptr, d_offsets_ptr, d_ids_ptr, ids_cap, d_needed_ptr=0, stream=None):
        """Device-resident id lists (bft_gpu_query_colors_dev): offsets (n + 1 uint64) and ids (uint32) in HBM, no synchronisation; when the
        ids number more than ids_cap only the first ids_cap of them are written (*d_needed_ptr says how many there are)."""
        _lib.check(self._lib.bft_gpu_query_colors_dev(self._h, C.c_void_p(d_kmers_ptr), n, C.c_void_p(d_bits_ptr), C.c_void_p(d_offsets_ptr),
                                                      C.c_void_p(d_ids_ptr or 0), ids_cap, C.c_void_p(d_needed_ptr or 0), C.c_void_p(stream or 0)))

    def query_color_rows_dev(self, d_kmers_ptr, n, d_bits_ptr, d_rows_ptr, d_scratch_u32_ptr, stream=None):
        """Device-resident colour rows (bft_gpu_query_color_rows_dev): n x CEIL(nb_genomes/8) bytes at d_rows_ptr, no synchronisation."""
        _lib.check(self._lib.bft_gpu_query_color_rows_dev(self._h, C.c_void_p(d_kmers_ptr), n, C.c_void_p(d_bits_ptr), C.c_void_p(d_rows_ptr),
                                                          C.c_void_p(d_scratch_u32_ptr), C.c_void_p(stream or 0)))

    def query_branching(self, kmers, with_counts=False):
        """-query_branching (src/file_io.c:897-1020): bit per k-mer, optionally (successors << 4) | predecessors."""
        kmers = self._chk(kmers)
        n = len(kmers)
        bits = np.zeros((n + 7) // 8, dtype=np.uint8)
        counts = np.zeros(n, dtype=np.uint8) if with_counts else None
        _lib.check(self._lib.bft_gpu_query_branching(self._h, kmers.ctypes.data, n, bits.ctypes.data,
                                                     counts.ctypes.data if with_counts else None))
        return (bits, counts) if with_counts else bits

    def query_branching_dev(self, d_kmers_ptr, n, d_bits_ptr, d_counts_ptr=None, stream=None):
        """Device-resident variant of query_branching: asynchronous on `stream`."""
        _lib.check(self._lib.bft_gpu_query_branching_dev(self._h, d_kmers_ptr, n, d_bits_ptr, d_counts_ptr, stream))

    def query_sequences(self, sequences, threshold, canonical=False):
        """query_sequence (include/bft.h:127) for a list of ASCII sequences: list of sorted genome-id lists."""
        enc = [x.encode() if isinstance(x, str) else bytes(x) for x in sequences]
        off = np.zeros(len(enc) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(e) for e in enc])
        blob = b"".join(enc) + b"\0"
        self.build()
        g = self.info()["genomes"]
        rows = np.zeros((len(enc), (g + 7) // 8), dtype=np.uint8)
        _lib.check(self._lib.bft_gpu_query_sequences(self._h, blob, off.ctypes.data, len(enc), float(threshold), int(canonical),
                                                     rows.ctypes.data))
        unp = np.unpackbits(rows, axis=1, bitorder="little")[:, :g] if g else np.zeros((len(enc), 0), np.uint8)
        return [np.flatnonzero(r).tolist() for r in unp]

    def query_sequences_dev(self, d_seqs_ptr, d_seq_off_ptr, n_seqs, total_chars, threshold, d_rows_ptr, canonical=False, stream=None):
        """Device-resident variant of query_sequences (raw pointers; rows of CEIL(genomes/8) bytes): asynchronous on `stream`."""
        _lib.check(self._lib.bft_gpu_query_sequences_dev(self._h, d_seqs_ptr, d_seq_off_ptr, n_seqs, total_chars, float(threshold), int(canonical),
                                                         d_rows_ptr, stream))

    def set_option(self, name, value):
        _lib.check(self._lib.bft_gpu_set_option(self._h, name.encode(), int(value)))

    # -- introspection ------------------------------------------------------------------------------------------
    def info(self):
        out = (C.c_uint64 * 16)()
        _lib.check(self._lib.bft_gpu_info(self._h, out, 16))
        return dict(zip(INFO_FIELDS, [int(x) for x in out]))

    def kernel_time(self, reset=True):
        ms, n = C.c_double(), C.c_uint64()
        _lib.check(self._lib.bft_gpu_kernel_time(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def build_time(self):
        out = (C.c_double * 25)()
        _lib.check(self._lib.bft_gpu_build_time(self._h, out, 25))
        return dict(zip(["gpu_sort_dedupe_ms", "color_intern_ms", "assemble_ms", "sort_redone_buckets", "derive_ms",
                         "query_wgs_per_cu", "tune_1wg_ms", "tune_2wg_ms", "query_probe_rows", "kmer_hash_lines", "kmer_hash_fill_ms",
                         "sort_max_bucket", "intern_exact_passes", "process_hipmalloc_ms", "root_tables", "tune_root_direct_ms", "tune_root_range_ms", "node_hash_keys", "node_hash_dropped", "tune_2wg768_ms",
                         "claims_static_launches", "kmer_hash_slots", "kmer_hash_dbits", "kmer_hash_maxd", "kmer_hash_overflow"], list(out)))

    def build_stages(self):
        """[(stage name, GPU ms, algorithmic bytes)] of the last build (set_option("build_stages", 1) before it)."""
        n = C.c_int()
        _lib.check(self._lib.bft_gpu_build_stages(self._h, None, 0, None, None, 0, C.byref(n)))
        if n.value == 0:
            return []
        names = C.create_string_buffer(256 * n.value)
        ms, by = (C.c_double * n.value)(), (C.c_double * n.value)()
        _lib.check(self._lib.bft_gpu_build_stages(self._h, names, len(names), ms, by, n.value, C.byref(n)))
        return list(zip(names.value.decode().split("\n")[:n.value], list(ms), list(by)))

    FOOTPRINT_FIELDS = ["kmer_table", "colorset_per_kmer", "colorset_dictionary", "containers", "flat_ccs", "root_tables", "node_prefix_hash", "kmer_hash",
                        "dictionary_bitmaps", "hash_table", "pair_store", "insertion_log"]

    def footprint(self):
        """Bytes in HBM per part of the handle (bft_gpu_footprint; src/printMemory.c:255 reports the reference's by container kind)."""
        out = (C.c_uint64 * 12)()
        _lib.check(self._lib.bft_gpu_footprint(self._h, out, 12))
        return dict(zip(self.FOOTPRINT_FIELDS, [int(x) for x in out]))

    def debug_array(self, name, dtype=np.uint8):
        n = C.c_uint64()
        _lib.check(self._lib.bft_gpu_debug_get_array(self._h, name.encode(), None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        _lib.check(self._lib.bft_gpu_debug_get_array(self._h, name.encode(), out.ctypes.data, n.value, C.byref(n)))
        return out.view(dtype)

    def extract(self):
        n = C.c_uint64()
        _lib.check(self._lib.bft_gpu_extract(self._h, None, None, 0, C.byref(n)))
        kmers = np.zeros((n.value, self.nb), dtype=np.uint8)
        cs = np.zeros(n.value, dtype=np.uint32)
        _lib.check(self._lib.bft_gpu_extract(self._h, kmers.ctypes.data, cs.ctypes.data, n.value, C.byref(n)))
        return kmers, cs

    def colorset(self, cs):
        n = C.c_uint32()
        _lib.check(self._lib.bft_gpu_colorset(self._h, int(cs), None, 0, C.byref(n)))
        ids = np.zeros(n.value, dtype=np.uint32)
        _lib.check(self._lib.bft_gpu_colorset(self._h, int(cs), ids.ctypes.data, n.value, C.byref(n)))
        return ids.tolist()

    def colorset_annot(self, cs):
        """The colour set as the reference's annotation bytes (BFT_annotation::annot, src/bft.c:363-387)."""
        n = C.c_uint32()
        _lib.check(self._lib.bft_gpu_colorset_annot(self._h, int(cs), None, 0, C.byref(n)))
        out = np.zeros(max(1, n.value), dtype=np.uint8)
        _lib.check(self._lib.bft_gpu_colorset_annot(self._h, int(cs), out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value].tobytes()


class BFTGroup:
    """One built index replicated on several GPUs of this process; host batches are sharded over them (bft_gpu_group_*)."""

    def __init__(self, bft, devices):
        self._lib = _lib.load()
        self._bft = bft
        arr = (C.c_int * len(devices))(*devices)
        g = C.c_void_p()
        _lib.check(self._lib.bft_gpu_group_create(bft._h, bft.device, arr, len(devices), C.byref(g)))
        self._g = g
        self.nb = bft.nb

    def size(self):
        return self._lib.bft_gpu_group_size(self._g)

    def close(self):
        if self._g:
            self._lib.bft_gpu_group_free(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def query_presence(self, kmers):
        kmers = self._bft._chk(kmers)
        bits = np.zeros((len(kmers) + 7) // 8, dtype=np.uint8)
        _lib.check(self._lib.bft_gpu_group_query_presence(self._g, kmers.ctypes.data, len(kmers), bits.ctypes.data))
        return bits

    def query_color_rows(self, kmers):
        kmers = self._bft._chk(kmers)
        n = len(kmers)
        rowbytes = (self._bft.info()["genomes"] + 7) // 8
        bits = np.zeros((n + 7) // 8, dtype=np.uint8)
        rows = np.zeros((n, rowbytes), dtype=np.uint8)
        _lib.check(self._lib.bft_gpu_group_query_color_rows(self._g, kmers.ctypes.data, n, bits.ctypes.data, rows.ctypes.data))
        return bits, rows

    def query_branching(self, kmers, with_counts=False):
        kmers = self._bft._chk(kmers)
        n = len(kmers)
        bits = np.zeros((n + 7) // 8, dtype=np.uint8)
        counts = np.zeros(n, dtype=np.uint8) if with_counts else None
        _lib.check(self._lib.bft_gpu_group_query_branching(self._g, kmers.ctypes.data, n, bits.ctypes.data, counts.ctypes.data if with_counts else None))
        return (bits, counts) if with_counts else bits

    # -- device-resident batches: one per slot, in the memory of that slot's GPU; only enqueues (bft_gpu_group_*_dev) --------------
    def member_device(self, i):
        return self._lib.bft_gpu_group_member_device(self._g, i)

    def member_footprint(self, i):
        out = (C.c_uint64 * 12)()
        _lib.check(self._lib.bft_gpu_group_member_footprint(self._g, i, out, 12))
        return dict(zip(BFT.FOOTPRINT_FIELDS, [int(x) for x in out]))

    def _ptrs(self, vals):
        # the C side indexes bft_gpu_group_size(g) entries of every array: a shorter list would be an out-of-bounds host read
        if len(vals) != self.size():
            raise ValueError(f"one entry per slot of the group expected ({self.size()}), got {len(vals)}")
        return (C.c_void_p * len(vals))(*[C.c_void_p(v) if v else None for v in vals])

    def _counts(self, n):
        if len(n) != self.size():
            raise ValueError(f"one batch size per slot of the group expected ({self.size()}), got {len(n)}")
        return (C.c_uint64 * len(n))(*n)

    def query_presence_dev(self, d_kmers, n, d_bits, streams=None):
        """d_kmers / d_bits / streams: device pointers (ints) per slot; n: k-mers per slot"""
        ns = self._counts(n)
        _lib.check(self._lib.bft_gpu_group_query_presence_dev(self._g, self._ptrs(d_kmers), ns, self._ptrs(d_bits), self._ptrs(streams) if streams else None))

    def query_color_rows_dev(self, d_kmers, n, d_bits, d_rows, d_scratch, streams=None):
        ns = self._counts(n)
        _lib.check(self._lib.bft_gpu_group_query_color_rows_dev(self._g, self._ptrs(d_kmers), ns, self._ptrs(d_bits), self._ptrs(d_rows), self._ptrs(d_scratch),
                                                                self._ptrs(streams) if streams else None))

    def query_branching_dev(self, d_kmers, n, d_bits, d_counts=None, streams=None):
        ns = self._counts(n)
        _lib.check(self._lib.bft_gpu_group_query_branching_dev(self._g, self._ptrs(d_kmers), ns, self._ptrs(d_bits), self._ptrs(d_counts) if d_counts else None,
                                                               self._ptrs(streams) if streams else None))


def shard(n, parts, i):
    """bft_gpu_group_shard: the contiguous 64-aligned slice [begin, end) of an n-query batch for part i of `parts`"""
    a, b = C.c_uint64(), C.c_uint64()
    _lib.check(_lib.load().bft_gpu_group_shard(n, parts, i, C.byref(a), C.byref(b)))
    return a.value, b.value


def cache_release():
    """bft_gpu_cache_release: the library's cache of released device blocks back to the HIP runtime; returns the bytes"""
    return int(_lib.load().bft_gpu_cache_release())


def create_cdbg(k, device=0):
    """create_cdbg(k, treshold_compression) (include/bft.h:62)."""
    return BFT(k, device)
