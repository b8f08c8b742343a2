"""The synthetic workloads of BASELINE.json / SURVEY.md 8d at their full sizes, generated on the GPU with torch (input
plumbing only: genomes, packed k-mer windows, query batches and the sorted key tables the ground-truth checks search).
Shared by bench.py, tests/test_gpu_configs.py and tools/bench_*.py so that every place measures and checks the same data.

 config 3: 100 genomes = one 2 Mbp ancestor with 1 % i.i.d. SNPs each, inserted genome by genome (ids ascending)
 config 4: the config-3 index, 10^9 / 8 presence queries per GPU (50 % stored k-mers, 50 % single-SNP mutants)
 config 5: k = 63, 2000 colours x 20 kbp variants of one ancestor; branching bits + colour rows

Packed k-mers use the reference's layout (src/fasta.c:3-53); a "key" is the packed k-mer read as little-endian int64
words (one column for k <= 32, two for k <= 64), which orders and compares k-mers exactly.
"""
import numpy as np


def pack_windows(codes, k):
    """codes: uint8 tensor [G] of nucleotide codes on a device -> all len-k windows, packed [G-k+1, CEIL(2k/8)]."""
    import torch
    n = codes.numel() - k + 1
    nb = (2 * k + 7) // 8
    win = codes.unfold(0, k, 1)
    pad = torch.zeros((n, nb * 4), dtype=torch.uint8, device=codes.device)
    pad[:, :k] = win
    q = pad.view(n, nb, 4)
    return (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).contiguous()


def keys_of(packed):
    """packed [n, nb] (nb <= 16) -> int64 keys: [n] for nb <= 8, else [n, 2] with the HIGH word (bytes 8..15) first, so
    that row-wise lexicographic order is a total order on k-mers."""
    import torch
    n, nb = packed.shape
    assert nb <= 16
    nw = 1 if nb <= 8 else 2
    pad = torch.zeros((n, 8 * nw), dtype=torch.uint8, device=packed.device)
    pad[:, :nb] = packed
    w = pad.view(torch.int64).reshape(n, nw)
    return w[:, 0].contiguous() if nw == 1 else torch.stack([w[:, 1], w[:, 0]], dim=1).contiguous()


def packed_of(keys, k):
    """inverse of keys_of for one-column keys"""
    import torch
    nb = (2 * k + 7) // 8
    return keys.contiguous().view(torch.uint8).reshape(-1, 8)[:, :nb].contiguous()


def unique_keys(keys):
    import torch
    return torch.unique(keys) if keys.dim() == 1 else torch.unique(keys, dim=0)


def member(sorted_keys, q):
    """Exact membership of q in the sorted, distinct key table (bool tensor)."""
    import torch
    if sorted_keys.dim() == 1:
        p = torch.searchsorted(sorted_keys, q).clamp(max=sorted_keys.numel() - 1)
        return sorted_keys[p] == q
    hi = sorted_keys[:, 0].contiguous()
    qh = q[:, 0].contiguous()
    lo_i = torch.searchsorted(hi, qh, right=False)
    hi_i = torch.searchsorted(hi, qh, right=True)
    width = int((hi_i - lo_i).max().item()) if q.shape[0] else 0  # k-mers sharing their high word: a handful at most
    out = torch.zeros(q.shape[0], dtype=torch.bool, device=q.device)
    last = sorted_keys.shape[0] - 1
    for j in range(width):
        p = lo_i + j
        ok = p < hi_i
        out |= ok & (sorted_keys[p.clamp(max=last), 1] == q[:, 1])
    return out


def snp_mutate_keys(qk, k, frac, gen):
    """Single-SNP mutants of a fraction `frac` of one-column keys (k <= 31: nucleotide j at bits 2j)."""
    import torch
    n = qk.numel()
    dev = qk.device
    mut = torch.rand(n, generator=gen, device=dev) < frac
    pos = torch.randint(0, k, (n,), generator=gen, device=dev)
    delta = torch.randint(1, 4, (n,), generator=gen, device=dev)
    nt = (qk >> (2 * pos)) & 3
    return torch.where(mut, (qk & ~(torch.full_like(qk, 3) << (2 * pos))) | (((nt + delta) & 3) << (2 * pos)), qk)


def snp_mutate_packed(q, k, frac, gen):
    """Single-SNP mutants of a fraction `frac` of packed k-mers [n, nb] (any k), in place on a copy."""
    import torch
    n = q.shape[0]
    dev = q.device
    q = q.clone()
    mut = torch.rand(n, generator=gen, device=dev) < frac
    pos = torch.randint(0, k, (n,), generator=gen, device=dev)
    delta = torch.randint(1, 4, (n,), generator=gen, device=dev).to(torch.uint8)
    byte = (pos // 4).long()
    sh = (2 * (pos % 4)).to(torch.uint8)
    rows = torch.arange(n, device=dev)
    cur = q[rows, byte]
    new = (((cur >> sh) & 3) + delta) & 3
    q[rows, byte] = torch.where(mut, (cur & ~(torch.full_like(cur, 3) << sh)) | (new << sh), cur)
    return q


def make_queries_on_device(union_kmers, k, n, seed, device):
    """Packed host k-mers [m, B] (any k) -> n device-resident queries: 50 % sampled as they are, 50 % single-SNP mutants (the
    config-2 batch of SURVEY.md 8d), built on the GPU in chunks."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    U = torch.from_numpy(union_kmers).to(device)
    out = torch.empty((n, U.shape[1]), dtype=torch.uint8, device=device)
    chunk = 1 << 24
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        idx = torch.randint(0, U.shape[0], (m,), generator=g, device=device)
        out[a:a + m] = snp_mutate_packed(U[idx], k, 0.5, g)
    return out


class PanGenome:
    """n_genomes variants (snp_rate i.i.d. SNPs) of one random ancestor of genome_len nt, generated on `device` from `seed`.
    genome(g) regenerates genome g deterministically (codes uint8 [genome_len])."""

    def __init__(self, n_genomes, genome_len, snp_rate, seed, device):
        import torch
        self.n, self.glen, self.rate, self.seed, self.dev = n_genomes, genome_len, snp_rate, seed, device
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.anc = torch.randint(0, 4, (genome_len,), generator=g, device=device, dtype=torch.uint8)

    def genome(self, gid):
        import torch
        g = torch.Generator(device=self.dev)
        g.manual_seed(self.seed * 1000003 + 17 * gid + 1)
        m = torch.rand(self.glen, generator=g, device=self.dev) < self.rate
        delta = torch.randint(1, 4, (self.glen,), generator=g, device=self.dev, dtype=torch.uint8)
        return torch.where(m, (self.anc + delta) & 3, self.anc)


def build_index(bft, pan, k, on_insert=None):
    """insertKmers genome by genome (ids ascending, device-resident batches), then the bulk build.  Returns the list of
    sorted distinct key tables, one per genome (the ground truth of presence and of colour sets) and the number of
    k-mer windows handed to the library."""
    import torch
    keys = []
    n_in = 0
    for gid in range(pan.n):
        packed = pack_windows(pan.genome(gid), k)
        if on_insert is not None:
            on_insert(gid, packed)
        else:
            # stream-ordered on torch's current stream: `packed` may go back to the caching allocator right away
            bft.insert_kmers_dev_async(packed.data_ptr(), packed.shape[0], gid, torch.cuda.current_stream().cuda_stream)
        n_in += packed.shape[0]
        keys.append(unique_keys(keys_of(packed)))
        del packed
    bft.build()
    return keys, n_in


def union_of(per_genome_keys):
    import torch
    return unique_keys(torch.cat(per_genome_keys))


def presence_batch(allk, k, nq, gen, mutant_frac=0.5):
    """nq queries drawn from the stored k-mers, a fraction turned into single-SNP mutants (near misses that walk deep
    into the trie); returns (packed device tensor [nq, B], one-column keys).  k <= 31."""
    import torch
    idx = torch.randint(0, allk.numel(), (nq,), generator=gen, device=allk.device)
    qk = snp_mutate_keys(allk[idx], k, mutant_frac, gen)
    return packed_of(qk, k), qk


def bits_to_bool(dbits, n):
    """device presence bitmap (uint8 tensor) -> bool tensor [n] on the same device"""
    import torch
    b = dbits[: (n + 7) // 8]
    sh = torch.arange(8, device=b.device, dtype=torch.uint8)
    return ((b[:, None] >> sh[None, :]) & 1).reshape(-1)[:n].bool()
