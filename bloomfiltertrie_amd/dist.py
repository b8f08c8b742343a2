"""Multi-GPU sharding of a query batch: one process per GPU, the trie image replicated in each
GPU's HBM, contiguous query slices per rank, one gather of the presence bitmaps (RCCL all_gather over xGMI on
GPUs; gloo in the CPU unit tests).  No exchange step exists between trie levels, so nothing else is collective.
"""
import numpy as np


def shard_bounds(n, world_size, rank, align=64):
    """Contiguous slice [a, b) of an n-query batch for `rank`; slice starts are multiples of `align` (64 queries =
    one u64 of the presence bitmap) so that per-rank bitmaps concatenate bytewise."""
    per = -(-n // world_size)
    per = -(-per // align) * align
    a = min(n, rank * per)
    b = min(n, a + per)
    return a, b, per


def gather_bitmaps(local_bits, n, world_size, rank, per, group=None):
    """all_gather the per-rank presence bitmaps (padded to `per` queries) and trim to CEIL(n/8) bytes.

    local_bits: torch uint8 tensor with per/8 bytes (device tensor under RCCL, CPU tensor under gloo)."""
    import torch
    import torch.distributed as dist
    nbytes = per // 8
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=local_bits.device)
    buf[: local_bits.numel()] = local_bits
    out = torch.empty(nbytes * world_size, dtype=torch.uint8, device=local_bits.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return out[: (n + 7) // 8]


def query_presence_sharded(bft, kmers, group=None):
    """Shard a host batch across the ranks of the default process group, query each slice on this rank's GPU and
    return the full bitmap on every rank.  Under RCCL the slice goes to the GPU once (one pinned, asynchronous copy), the
    query runs device-resident (bft_gpu_query_presence_dev) and its bitmap is gathered straight from HBM; only the
    gathered result comes back to the host.  Under gloo (CPU unit tests, no GPU) the handle's host entry point is used."""
    import torch
    import torch.distributed as dist
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    n = len(kmers)
    a, b, per = shard_bounds(n, ws, rk)
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", bft.device)
        local = torch.zeros(per // 8, dtype=torch.uint8, device=dev)
        if b > a:
            host = torch.from_numpy(np.ascontiguousarray(kmers[a:b]))
            dq = host.pin_memory().to(dev, non_blocking=True)
            bft.query_presence_dev(dq.data_ptr(), b - a, local.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        return gather_bitmaps(local, n, ws, rk, per, group).cpu().numpy()
    bits = bft.query_presence(kmers[a:b]) if b > a else np.zeros(0, np.uint8)
    t = torch.from_numpy(np.ascontiguousarray(bits))
    return gather_bitmaps(t, n, ws, rk, per, group).cpu().numpy()


def gather_rows(local_rows, n, world_size, per, width, group=None):
    """all_gather fixed-width per-query rows (`per` rows of `width` bytes from every rank) and trim to n rows."""
    import torch
    import torch.distributed as dist
    buf = torch.zeros(per * width, dtype=torch.uint8, device=local_rows.device)
    buf[: local_rows.numel()] = local_rows.reshape(-1)
    out = torch.empty(per * width * world_size, dtype=torch.uint8, device=local_rows.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return out[: n * width].reshape(n, width)


def query_color_rows_sharded(bft, kmers, nb_genomes, group=None):
    """The colour-row query (-query_kmers with its CSV rows, src/file_io.c:726-768) sharded like query_presence_sharded: every rank
    answers its slice -- presence bitmap and one CEIL(nb_genomes/8)-byte genome bitmap per k-mer --, both are gathered (fixed-width
    rows concatenate like the bitmaps do: SURVEY.md 8e) and every rank returns (bits, rows[n, width])."""
    import torch
    import torch.distributed as dist
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    n, width = len(kmers), (int(nb_genomes) + 7) // 8
    a, b, per = shard_bounds(n, ws, rk)
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", bft.device)
        local = torch.zeros(per // 8, dtype=torch.uint8, device=dev)
        rows = torch.zeros(per * width, dtype=torch.uint8, device=dev)
        if b > a:
            dq = torch.from_numpy(np.ascontiguousarray(kmers[a:b])).pin_memory().to(dev, non_blocking=True)
            scratch = torch.empty(b - a, dtype=torch.int32, device=dev)
            bft.query_color_rows_dev(dq.data_ptr(), b - a, local.data_ptr(), rows.data_ptr(), scratch.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    else:
        lb, lr = bft.query_color_rows(kmers[a:b]) if b > a else (np.zeros(0, np.uint8), np.zeros((0, width), np.uint8))
        local, rows = torch.from_numpy(np.ascontiguousarray(lb)), torch.from_numpy(np.ascontiguousarray(lr).reshape(-1))
    bits = gather_bitmaps(local, n, ws, rk, per, group)
    return bits.cpu().numpy(), gather_rows(rows, n, ws, per, width, group).cpu().numpy()


def query_branching_sharded(bft, kmers, group=None):
    """-query_branching (src/file_io.c:897-1020) sharded the same way: (branching bitmap, (successors << 4 | predecessors) per k-mer)
    on every rank."""
    import torch
    import torch.distributed as dist
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    n = len(kmers)
    a, b, per = shard_bounds(n, ws, rk)
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", bft.device)
        local = torch.zeros(per // 8, dtype=torch.uint8, device=dev)
        counts = torch.zeros(per, dtype=torch.uint8, device=dev)
        if b > a:
            dq = torch.from_numpy(np.ascontiguousarray(kmers[a:b])).pin_memory().to(dev, non_blocking=True)
            bft.query_branching_dev(dq.data_ptr(), b - a, local.data_ptr(), counts.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    else:
        lb, lc = bft.query_branching(kmers[a:b], with_counts=True) if b > a else (np.zeros(0, np.uint8), np.zeros(0, np.uint8))
        local, counts = torch.from_numpy(np.ascontiguousarray(lb)), torch.from_numpy(np.ascontiguousarray(lc))
    bits = gather_bitmaps(local, n, ws, rk, per, group)
    return bits.cpu().numpy(), gather_rows(counts, n, ws, per, 1, group).cpu().numpy().reshape(-1)


class GatherPipeline:
    """The step / drain logic of bench.py's multi-GPU loop: every step answers this rank's resident shard into one of two
    result buffers and starts the all_gather of that buffer asynchronously (RCCL runs it on its own stream), so the
    gather of step i overlaps the query kernel of step i+1; a buffer is only reused once the gather that read it has
    completed.  `query(buf)` must fill the uint8 tensor `buf` (the presence bitmap of the local shard) on the current
    stream.  With use_dist False there is one buffer and no collective."""

    def __init__(self, query, nbytes_local, world, device, use_dist, group=None):
        import torch
        self.query, self.world, self.use_dist, self.group = query, world, use_dist, group
        self.nbuf = 2 if use_dist else 1
        self.bits = [torch.zeros(nbytes_local, dtype=torch.uint8, device=device) for _ in range(self.nbuf)]
        self.gathered = [torch.empty(nbytes_local * world, dtype=torch.uint8, device=device) for _ in range(self.nbuf)] if use_dist else None
        self.pending = [None] * self.nbuf
        self.steps = 0

    def step(self):
        import torch.distributed as dist
        b = self.steps % self.nbuf
        self.steps += 1
        if self.pending[b] is not None:
            self.pending[b].wait()  # the gather that last read this buffer
            self.pending[b] = None
        self.query(self.bits[b])
        if self.use_dist:
            self.pending[b] = dist.all_gather_into_tensor(self.gathered[b], self.bits[b], group=self.group, async_op=True)

    def drain(self):
        for b in range(self.nbuf):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None

    def last(self):
        """(local bitmap, gathered bitmaps or None) of the most recent step"""
        b = (self.steps - 1) % self.nbuf
        return self.bits[b], (self.gathered[b] if self.use_dist else None)


def replicate_image(bft, device, src=0, group=None, always_copy=False):
    """Replicate the index built on rank `src` into the HBM of every rank's GPU with ONE broadcast (RCCL over
    xGMI): `bft` is the built index on rank src and ignored (may be None) elsewhere; returns a BFT on `device`
    on every rank (rank src gets its own object back unless always_copy).  SURVEY.md 8e: "trie image replicated
    per GPU"."""
    import torch
    import torch.distributed as dist
    from .bft import BFT
    rk = dist.get_rank(group)
    dev = torch.device("cuda", device)
    size = torch.zeros(1, dtype=torch.int64, device=dev)
    if rk == src:
        size[0] = bft.image_size()
    dist.broadcast(size, src, group=group)
    nbytes = int(size.item())
    blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if rk == src:
        bft.image_pack(blob.data_ptr(), nbytes, torch.cuda.current_stream(dev).cuda_stream)
    dist.broadcast(blob, src, group=group)
    if rk == src and not always_copy:
        return bft
    torch.cuda.synchronize(dev)
    return BFT.from_image(blob.data_ptr(), nbytes, device=device)
