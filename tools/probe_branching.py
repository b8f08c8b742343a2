#!/usr/bin/env python3
"""k_branching under both residencies on the perf_probe workloads (tuning aid)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, synth as S  # noqa: E402
from bloomfiltertrie_amd.workloads import make_queries_on_device  # noqa: E402
from tools.perf_probe import workload  # noqa: E402

nq = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
dev = torch.device("cuda", 0)
for wl in (sys.argv[1] if len(sys.argv) > 1 else "cfg2,deep2,k63").split(","):
    k, gk = workload(wl)
    t = BFT(k)
    for g, km in enumerate(gk):
        t.insert_kmers(km, g)
    t.build()
    union = S.distinct(np.concatenate(gk)) if len(gk) > 1 else gk[0]
    dq = make_queries_on_device(union, k, nq, 5, dev)
    bits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    res = {"workload": wl, "auto": t.build_time()["query_wgs_per_cu"]}
    ref = None
    for wg in (1, 2):
        t.set_option("query_wgs_per_cu", wg)
        t.query_branching_dev(dq.data_ptr(), nq, bits.data_ptr(), None, st)
        torch.cuda.synchronize()
        t.kernel_time(reset=True)
        for _ in range(3):
            t.query_branching_dev(dq.data_ptr(), nq, bits.data_ptr(), None, st)
        torch.cuda.synchronize()
        ms, n = t.kernel_time(reset=True)
        res[f"wgs{wg}_M_kmers_per_s"] = round(nq / (ms / n) / 1e3, 1)
        cur = bits.clone()
        res["same"] = True if ref is None else bool(torch.equal(ref, cur))
        ref = cur
    print(json.dumps(res), flush=True)
