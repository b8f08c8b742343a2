"""bloomfiltertrie_amd -- MI355X-native batched k-mer presence / colour / insertion path of the Bloom Filter Trie.

The compute path is the HIP library csrc/libbft_gpu.so (C-ABI: include/bft_gpu.h); this package is the
thin host-side mirror of the reference interface plus synthetic-data helpers.  No CPU fallback exists.
"""
from . import synth  # noqa: F401


def __getattr__(name):
    if name in ("BFT", "BFTGroup", "create_cdbg", "shard", "cache_release"):
        from . import bft
        return getattr(bft, name)
    raise AttributeError(name)
