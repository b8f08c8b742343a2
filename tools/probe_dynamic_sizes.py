#!/usr/bin/env python3
"""From which batch size on do claimed chunks beat the static split of k_query_kh ("query_dynamic_min")?  usage: probe_dynamic_sizes.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W  # noqa: E402

k, nq = 27, 1 << 26
dev = torch.device("cuda", 0)
pan = W.PanGenome(100, 2_000_000, 0.01, 4242, dev)
t = BFT(k)
keys, _ = W.build_index(t, pan, k)
allk = W.union_of(keys)
del keys
g = torch.Generator(device=dev)
g.manual_seed(99)
dq, qk = W.presence_batch(allk, k, nq, g)
bits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for lg in (18, 19, 20, 21, 22, 23, 24, 25, 26):
    m = 1 << lg
    out = {}
    for rep in range(2):
        for name, dmin in (("static_us", 1 << 40), ("claimed_us", 0)):
            t.set_option("query_dynamic_min", dmin)
            for _ in range(3):
                t.query_presence_dev(dq.data_ptr(), m, bits.data_ptr(), stream)
            torch.cuda.synchronize()
            t.kernel_time(reset=True)
            reps = max(10, min(200, (1 << 27) // m))
            for _ in range(reps):
                t.query_presence_dev(dq.data_ptr(), m, bits.data_ptr(), stream)
            torch.cuda.synchronize()
            ms, n = t.kernel_time(reset=True)
            out.setdefault(name, []).append(round(1000 * ms / n, 2))
    print(json.dumps({"log2_queries": lg, **out}), flush=True)
