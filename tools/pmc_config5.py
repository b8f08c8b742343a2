#!/usr/bin/env python3
"""Workload for rocprofv3 passes over the config-5 kernels (k=63, 2000 colours): build the index (workloads.py), then `reps`
launches each of k_branching (10^7 k-mers) and of the colour-row path (k_color_rows_kh: lookup and rows in one launch, 4x10^6 k-mers).
usage: pmc_config5.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, _lib as L, workloads as W  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
k, G, glen, nq, nqc = 63, 2000, 20000, 10_000_000, 4_000_000
pan = W.PanGenome(G, glen, 0.01, 77, dev)
t = BFT(k)
keys, _ = W.build_index(t, pan, k)
allk = W.union_of(keys)
del keys
g = torch.Generator(device=dev)
g.manual_seed(1)
idx = torch.randint(0, allk.shape[0], (nq,), generator=g, device=dev)
sel = allk[idx]
stored = torch.cat([sel[:, 1:2], sel[:, 0:1]], dim=1).contiguous().view(torch.uint8).reshape(nq, 16)
dq = W.snp_mutate_packed(stored, k, 0.5, g)
stream = torch.cuda.current_stream().cuda_stream
dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
rowbytes = (G + 7) // 8
drows = torch.zeros((nqc, rowbytes), dtype=torch.uint8, device=dev)
dscr = torch.zeros(nqc, dtype=torch.int32, device=dev)
for _ in range(reps):
    L.check(t._lib.bft_gpu_query_branching_dev(t._h, dq.data_ptr(), nq, dbits.data_ptr(), None, stream))
for _ in range(reps):
    L.check(t._lib.bft_gpu_query_color_rows_dev(t._h, dq.data_ptr(), nqc, dbits.data_ptr(), drows.data_ptr(), dscr.data_ptr(), stream))
torch.cuda.synchronize()
print("done", t.info()["image_bytes"])
