#!/usr/bin/env python3
"""Colour rows (presence + CEIL(G/8)-byte genome bitmap per k-mer) through bft_gpu_query_color_rows_dev on the config-2 / config-4
indexes (10 / 100 genomes), device-resident, every row of a slice checked against the inserting genomes.
usage: bench_color_rows.py [cfg2|cfg4] [queries]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W, _lib as L  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
genomes = 10 if wl == "cfg2" else 100
nq = int(sys.argv[2]) if len(sys.argv) > 2 else (100_000_000 if wl == "cfg2" else 50_000_000)
k = 27
dev = torch.device("cuda", 0)
pan = W.PanGenome(genomes, 2_000_000, 0.01, 4242, dev)
t = BFT(k)
keys, _ = W.build_index(t, pan, k)
allk = W.union_of(keys)
g = torch.Generator(device=dev)
g.manual_seed(99)
dq, qk = W.presence_batch(allk, k, nq, g)
rb = (genomes + 7) // 8
bits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
rows = torch.zeros((nq, rb), dtype=torch.uint8, device=dev)
scr = torch.zeros(nq, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
call = lambda: L.check(t._lib.bft_gpu_query_color_rows_dev(t._h, dq.data_ptr(), nq, bits.data_ptr(), rows.data_ptr(), scr.data_ptr(), st))
call(); call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    call()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
ns = 2_000_000
got = torch.from_numpy(np.unpackbits(rows[:ns].cpu().numpy(), axis=1, bitorder="little")[:, :genomes].astype(bool)).to(dev)
exp = torch.stack([W.member(keys[i], qk[:ns]) for i in range(genomes)], dim=1)
print(json.dumps({"workload": wl, "genomes": genomes, "queries": nq, "row_bytes": rb, "ms": round(ms, 3), "G_kmers_per_s": round(nq / ms / 1e6, 2),
                  "rows_of_slice_ok": bool((got == exp).all()), "slice": ns}))
