#!/usr/bin/env python3
"""Launch-by-launch durations of k_query_kh (claimed chunks / static split) on the 100-genome index: is the time of a launch stable
inside a process?  usage: probe_dynamic_series.py [launches]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
k, nq = 27, 125_000_000
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
for mode, chunk in ((1, 8), (0, 8), (1, 4), (1, 16), (1, 32), (1, 64), (1, 256)):
    t.set_option("query_dynamic", mode)
    t.set_option("query_chunk", chunk)
    for back_to_back in (True,):
        t.query_presence_dev(dq.data_ptr(), nq, bits.data_ptr(), stream)
        torch.cuda.synchronize()
        t.kernel_time(reset=True)
        series = []
        if back_to_back:  # enqueued together, one synchronisation: the mean only
            for _ in range(reps):
                t.query_presence_dev(dq.data_ptr(), nq, bits.data_ptr(), stream)
            torch.cuda.synchronize()
            ms, n = t.kernel_time(reset=True)
            series = [round(ms / n, 4)]
        else:  # one at a time, the GPU idle in between
            for _ in range(reps):
                t.query_presence_dev(dq.data_ptr(), nq, bits.data_ptr(), stream)
                torch.cuda.synchronize()
                ms, n = t.kernel_time(reset=True)
                series.append(round(ms / n, 3))
        print(json.dumps({"dynamic": mode, "chunk": chunk, "back_to_back": back_to_back, "ms": series}), flush=True)
