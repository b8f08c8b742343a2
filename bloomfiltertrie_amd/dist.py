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
