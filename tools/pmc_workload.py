#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes: config-2 trie, a resident batch, a few k_query launches (no timing claims)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, synth as S  # noqa: E402
from bloomfiltertrie_amd.workloads import make_queries_on_device  # noqa: E402
from tools.perf_probe import workload  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
stop = int(os.environ.get("BFT_DEBUG_STOP", "0"))  # truncate the walk after a stage (tools/perf_probe.py --stops)
k, gk = workload(wl)
t = BFT(k)
for g, km in enumerate(gk):
    t.insert_kmers(km, g)
t.build()
union = S.distinct(np.concatenate(gk)) if len(gk) > 1 else gk[0]
dev = torch.device("cuda", 0)
dq = make_queries_on_device(union, k, nq, 99, dev)
dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
if stop:  # needs BFT_GPU_LIB=bloomfiltertrie_amd/csrc/libbft_gpu_probe.so (make -C bloomfiltertrie_amd/csrc probe)
    t.set_option("debug_stop", stop)
for _ in range(reps):
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("done", wl, nq, t.info()["image_bytes"])
