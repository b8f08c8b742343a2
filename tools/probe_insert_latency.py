#!/usr/bin/env python3
"""Development probe (GPU box): host-side cost of bft_gpu_insert_kmers_dev per call, for several batch sizes."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT  # noqa: E402

dev = torch.device("cuda", 0)
for n in (1000, 100_000, 2_000_000):
    t = BFT(27)
    buf = torch.randint(0, 256, (n, 7), dtype=torch.uint8, device=dev)
    buf[:, 6] &= 0x3F
    torch.cuda.synchronize()
    times = []
    for g in range(60):
        t0 = time.perf_counter()
        t.insert_kmers_dev(buf.data_ptr(), n, g)
        times.append(time.perf_counter() - t0)
    times_us = [round(x * 1e6, 1) for x in times]
    print(json.dumps({"n": n, "first_us": times_us[:4], "median_us": sorted(times_us)[30], "max_us": max(times_us), "sum_ms": round(sum(times) * 1e3, 2)}))
    t.close()
