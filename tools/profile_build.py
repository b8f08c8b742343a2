#!/usr/bin/env python3
"""Workload for rocprofv3 --kernel-trace --stats of the bulk build (config 3 at 20 or 100 genomes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT  # noqa: E402
from tools.bench_insert import pack_windows  # noqa: E402

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(4242)
glen, k = 2_000_000, 27
anc = torch.randint(0, 4, (glen,), generator=g, device=dev, dtype=torch.uint8)
t = BFT(k)
for gid in range(ng):
    m = torch.rand(glen, generator=g, device=dev) < 0.01
    delta = torch.randint(1, 4, (glen,), generator=g, device=dev, dtype=torch.uint8)
    packed = pack_windows(torch.where(m, (anc + delta) & 3, anc), k)
    t.insert_kmers_dev(packed.data_ptr(), packed.shape[0], gid)
    del packed
t.build()
print(t.info(), t.build_time())
