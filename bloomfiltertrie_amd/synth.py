"""Seeded synthetic genomes / k-mer batches in the reference's packed layout.

Packed k-mer layout = parseKmerCount (reference src/fasta.c:3-53, README.md:171-172):
nucleotide j -> byte j//4, bits 2*(j%4)..2*(j%4)+1, A=0 C=1 G=2 T=3; CEIL(2k/8) bytes per k-mer.
"""
import numpy as np

_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)
_CODE = np.full(256, 255, dtype=np.uint8)
for _c, _v in ((b"A", 0), (b"C", 1), (b"G", 2), (b"T", 3), (b"U", 3), (b"a", 0), (b"c", 1), (b"g", 2), (b"t", 3), (b"u", 3)):
    _CODE[_c[0]] = _v


def kmer_bytes(k: int) -> int:
    return (2 * k + 7) // 8


def random_genome(length: int, seed: int) -> np.ndarray:
    """Uniform ACGT codes (uint8 in 0..3)."""
    return np.random.default_rng(seed).integers(0, 4, size=length, dtype=np.uint8)


def mutate(genome: np.ndarray, rate: float, seed: int) -> np.ndarray:
    """i.i.d. SNPs: each position replaced by a different nucleotide with probability `rate`."""
    rng = np.random.default_rng(seed)
    g = genome.copy()
    m = rng.random(len(g)) < rate
    g[m] = (g[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
    return g


def pack_codes(codes: np.ndarray) -> np.ndarray:
    """[n, k] nucleotide codes -> [n, CEIL(2k/8)] packed bytes."""
    codes = np.asarray(codes, dtype=np.uint8)
    n, k = codes.shape
    nb = kmer_bytes(k)
    pad = np.zeros((n, nb * 4), dtype=np.uint8)
    pad[:, :k] = codes
    q = pad.reshape(n, nb, 4)
    return (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8)


def unpack_codes(packed: np.ndarray, k: int) -> np.ndarray:
    packed = np.asarray(packed, dtype=np.uint8)
    n, nb = packed.shape
    out = np.empty((n, nb, 4), dtype=np.uint8)
    for j in range(4):
        out[:, :, j] = (packed >> (2 * j)) & 3
    return out.reshape(n, nb * 4)[:, :k]


def kmers_of(genome: np.ndarray, k: int, chunk: int = 1 << 20) -> np.ndarray:
    """All len-k windows of a genome (in order), packed."""
    n = len(genome) - k + 1
    if n <= 0:
        return np.zeros((0, kmer_bytes(k)), dtype=np.uint8)
    win = np.lib.stride_tricks.sliding_window_view(genome, k)
    out = np.empty((n, kmer_bytes(k)), dtype=np.uint8)
    for a in range(0, n, chunk):
        out[a:a + chunk] = pack_codes(win[a:a + chunk])
    return out


def ascii_to_packed(seqs, k: int):
    """List of ASCII k-mers -> (packed [n, B], valid [n]); invalid characters give an all-zero k-mer
    and valid=False, as the reference CLI does (src/file_io.c:844-850)."""
    n = len(seqs)
    arr = np.zeros((n, k), dtype=np.uint8)
    valid = np.ones(n, dtype=bool)
    for i, s in enumerate(seqs):
        b = np.frombuffer(s.encode() if isinstance(s, str) else s, dtype=np.uint8)
        if len(b) < k:
            valid[i] = False
            continue
        c = _CODE[b[:k]]
        if (c == 255).any():
            valid[i] = False
        else:
            arr[i] = c
    return pack_codes(arr), valid


def packed_to_ascii(packed: np.ndarray, k: int):
    codes = unpack_codes(packed, k)
    return [bytes(_ASCII[row]).decode() for row in codes]


def row_keys(packed: np.ndarray) -> np.ndarray:
    """A 1-D array of sortable, hashable keys (one per packed row) for set operations."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n, nb = packed.shape
    if nb <= 8:
        pad = np.zeros((n, 8), dtype=np.uint8)
        pad[:, :nb] = packed
        return pad.view(np.uint64).reshape(n)
    return packed.view(np.dtype((np.void, nb))).reshape(n)


def distinct(packed: np.ndarray) -> np.ndarray:
    """Distinct rows in order of first occurrence."""
    keys = row_keys(packed)
    _, idx = np.unique(keys, return_index=True)
    idx.sort()
    return packed[idx]


def member(packed_queries: np.ndarray, packed_set: np.ndarray) -> np.ndarray:
    """Ground-truth set membership (bool per query)."""
    return np.isin(row_keys(packed_queries), row_keys(packed_set))


def to_bits(flags: np.ndarray) -> np.ndarray:
    """bool[n] -> presence bitmap, bit i%8 of byte i//8 (LSB first)."""
    return np.packbits(np.asarray(flags, dtype=bool), bitorder="little")


def from_bits(bits: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(np.asarray(bits, dtype=np.uint8), bitorder="little")[:n].astype(bool)


def snp_mutants(packed: np.ndarray, k: int, seed: int) -> np.ndarray:
    """Single-SNP mutants of packed k-mers (share prefixes with present k-mers: exercise deep paths)."""
    rng = np.random.default_rng(seed)
    out = packed.copy()
    n = len(out)
    pos = rng.integers(0, k, size=n)
    delta = rng.integers(1, 4, size=n).astype(np.uint8)
    byte = pos // 4
    sh = (2 * (pos % 4)).astype(np.uint8)
    cur = (out[np.arange(n), byte] >> sh) & 3
    new = (cur + delta) & 3
    out[np.arange(n), byte] = (out[np.arange(n), byte] & ~(np.uint8(3) << sh)) | (new << sh)
    return out


def low_entropy_kmers(n: int, k: int, n_prefixes: int, seed: int, levels: int = 1) -> np.ndarray:
    """k-mers drawn under few distinct 9-mer prefixes (per level) so that trie levels 2..L are exercised
    (SURVEY.md 8d 'Depth caveat')."""
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=(n, k), dtype=np.uint8)
    for lv in range(levels):
        pref = rng.integers(0, 4, size=(n_prefixes, 9), dtype=np.uint8)
        pick = rng.integers(0, n_prefixes, size=n)
        codes[:, 9 * lv:9 * lv + 9] = pref[pick]
    return distinct(pack_codes(codes))
