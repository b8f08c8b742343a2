#!/usr/bin/env python3
"""k_query_kh with its blocks dealt out by workgroup number against claimed chunks ("query_dynamic"), on the same table, in the placements
one process meets when the table is re-derived into fresh memory (BFT_GPU_POOL_MAX_MB=0).  usage: probe_dynamic.py [derivations] [k]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W  # noqa: E402

n_der = int(sys.argv[1]) if len(sys.argv) > 1 else 6
k = int(sys.argv[2]) if len(sys.argv) > 2 else 27
nq = 125_000_000
dev = torch.device("cuda", 0)
pan = W.PanGenome(100, 2_000_000, 0.01, 4242, dev)
t = BFT(k)
keys, _ = W.build_index(t, pan, k)
allk = W.union_of(keys)
del keys
g = torch.Generator(device=dev)
g.manual_seed(99)
dq, qk = W.presence_batch(allk, k, nq, g)
bits = [torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev) for _ in range(2)]
stream = torch.cuda.current_stream().cuda_stream
held = []


def timed(mode):
    t.set_option("query_dynamic", mode)
    t.query_presence_dev(dq.data_ptr(), nq, bits[mode].data_ptr(), stream)
    torch.cuda.synchronize()
    t.kernel_time(reset=True)
    for _ in range(10):
        t.query_presence_dev(dq.data_ptr(), nq, bits[mode].data_ptr(), stream)
    torch.cuda.synchronize()
    ms, n = t.kernel_time(reset=True)
    return round(ms / n, 4)


for d in range(n_der):
    a, b, a2 = timed(0), timed(1), timed(0)
    rec = {"derivation": d, "static_ms": [a, a2], "dynamic_ms": b, "same_bits": bool(torch.equal(bits[0], bits[1]))}
    if d < 2:  # blocks of 256 k-mers per claim
        for ch in (1, 2, 4, 16, 32, 64):
            t.set_option("query_chunk", ch)
            rec[f"dynamic_chunk{ch}_ms"] = timed(1)
        t.set_option("query_chunk", 4)
    print(json.dumps(rec), flush=True)
    t.set_option("kmer_hash", 0)
    held.append(torch.empty((97 + 61 * d) << 20, dtype=torch.uint8, device=dev))
    t.set_option("kmer_hash", 1)

# smaller batches: where the claims stop paying (the library takes the static split below 2^20 k-mers)
for m in (1 << 20, 1 << 22, 1 << 24):
    out = {}
    for mode in (0, 1):
        t.set_option("query_dynamic", mode)
        for _ in range(3):
            t.query_presence_dev(dq.data_ptr(), m, bits[mode].data_ptr(), stream)
        torch.cuda.synchronize()
        t.kernel_time(reset=True)
        for _ in range(50):
            t.query_presence_dev(dq.data_ptr(), m, bits[mode].data_ptr(), stream)
        torch.cuda.synchronize()
        ms, n = t.kernel_time(reset=True)
        out["dynamic_us" if mode else "static_us"] = round(1000 * ms / n, 2)
    print(json.dumps({"queries": m, **out}), flush=True)
