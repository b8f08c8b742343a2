#!/usr/bin/env python3
"""One launch-time figure of the headline kernel on the config-4 share with whatever library BFT_GPU_LIB names (A/B of two builds on one box).
usage: [BFT_GPU_LIB=...] probe_ab.py [k] [opt=value ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 27
dev = torch.device("cuda", 0)
pan = W.PanGenome(100, 2_000_000, 0.01, 4242, dev)
t = BFT(k)
for a in sys.argv[2:]:
    o, v = a.split("=")
    t.set_option(o, int(v))
keys, _ = W.build_index(t, pan, k)
allk = W.union_of(keys)
del keys
g = torch.Generator(device=dev)
g.manual_seed(99)
nq = 125_000_000
dq, qk = W.presence_batch(allk, k, nq, g)
truth = W.member(allk, qk)
dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
out = []
for rep in range(3):
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    ok = bool((W.bits_to_bool(dbits, nq) == truth).all())
    t.kernel_time(reset=True)
    for _ in range(10):
        t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), stream)
    torch.cuda.synchronize()
    ms, n = t.kernel_time(reset=True)
    out.append(round(nq / (ms / n) / 1e6, 2))
print(json.dumps({"lib": os.environ.get("BFT_GPU_LIB", "default"), "k": k, "opts": sys.argv[2:], "G_kmers_per_s": out, "ok": ok}), flush=True)
