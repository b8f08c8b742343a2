#!/usr/bin/env python3
"""With claimed chunks the launch time no longer follows the table's placement but still differs between processes (2.60 / 2.82 ms):
does it follow the placement of the query batch or of the output bitmap?  Copies of both inside one process.  usage: probe_dynamic_inputs.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W  # noqa: E402

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


def timed(q, b):
    t.query_presence_dev(q.data_ptr(), nq, b.data_ptr(), stream)
    torch.cuda.synchronize()
    t.kernel_time(reset=True)
    for _ in range(10):
        t.query_presence_dev(q.data_ptr(), nq, b.data_ptr(), stream)
    torch.cuda.synchronize()
    ms, n = t.kernel_time(reset=True)
    return round(ms / n, 4)


out = {"orig": timed(dq, bits), "dq_ptr_mod_2MiB": dq.data_ptr() % (2 << 20)}
held = []
for i in range(5):
    held.append(torch.empty((53 + 97 * i) << 20, dtype=torch.uint8, device=dev))
    q2 = dq.clone()
    out[f"dq_copy{i}"] = timed(q2, bits)
    held.append(q2)
for i in range(3):
    b2 = torch.zeros_like(bits)
    out[f"bits_copy{i}"] = timed(dq, b2)
    held.append(b2)
t.set_option("kmer_hash", 0)
t.set_option("kmer_hash", 1)
out["table_rederived"] = timed(dq, bits)
out["orig_again"] = timed(dq, bits)
print(json.dumps(out))
