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
    return the full bitmap on every rank."""
    import torch
    import torch.distributed as dist
    ws, rk = dist.get_world_size(group), dist.get_rank(group)
    n = len(kmers)
    a, b, per = shard_bounds(n, ws, rk)
    bits = bft.query_presence(kmers[a:b]) if b > a else np.zeros(0, np.uint8)
    dev = torch.device("cuda", bft.device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.from_numpy(np.ascontiguousarray(bits)).to(dev)
    return gather_bitmaps(t, n, ws, rk, per, group).cpu().numpy()


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
