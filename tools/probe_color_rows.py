#!/usr/bin/env python3
"""The colour-row kernel alone on config 5's index (2000 colours: 250-byte rows): bft_gpu_query_color_rows_dev timed by kernel_time is not
available for it, so the call is timed with events around a loop.  usage: probe_color_rows.py [queries]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, synth as S, _lib as L  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
k, G, glen = 63, 2000, 20000
anc = S.random_genome(glen, 77)
t = BFT(k)
gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 5000 + g), k)) for g in range(G)]
for g, km in enumerate(gk):
    t.insert_kmers(km, g)
t.build()
union = S.distinct(np.concatenate(gk))
rng = np.random.default_rng(1)
q = union[rng.integers(0, len(union), nq)]
q[nq // 2:] = S.snp_mutants(q[nq // 2:], k, 9)
q = np.ascontiguousarray(q[rng.permutation(nq)])
dev = torch.device("cuda", 0)
dq = torch.from_numpy(q).to(dev)
rb = (G + 7) // 8
bits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
rows = torch.zeros((nq, rb), dtype=torch.uint8, device=dev)
scr = torch.zeros(nq, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
call = lambda: L.check(t._lib.bft_gpu_query_color_rows_dev(t._h, dq.data_ptr(), nq, bits.data_ptr(), rows.data_ptr(), scr.data_ptr(), st))
pres = lambda: L.check(t._lib.bft_gpu_query_presence_dev(t._h, dq.data_ptr(), nq, bits.data_ptr(), st))
def timed(f, reps=10):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms_all, ms_p = timed(call), timed(pres)
print(json.dumps({"mode": os.environ.get("BFT_CR_MODE", "0"), "queries": nq, "row_bytes": rb, "call_ms": round(ms_all, 4), "presence_ms": round(ms_p, 4),
                  "rows_kernel_ms_est": round(ms_all - ms_p, 4), "GB_per_s_rows_kernel_est": round(nq * rb / (ms_all - ms_p) / 1e6, 1)}))
