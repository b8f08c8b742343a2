#!/usr/bin/env python3
"""Id-list colours (get_annotation + get_list_id_genomes, src/bft.c:363-387, 622-641) on a resident batch through bft_gpu_query_colors_dev:
presence bits, offsets and genome ids in HBM, no host round trip -- config-4 index (100 genomes) or config 2 (10), every list of a slice
checked against the inserting genomes, the image's footprint before and after (the sorted table must not come back).
usage: bench_color_lists.py [cfg2|cfg4] [queries]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
genomes = 10 if wl == "cfg2" else 100
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
k = 27
dev = torch.device("cuda", 0)
pan = W.PanGenome(genomes, 2_000_000, 0.01, 4242, dev)
t = BFT(k)
keys, _ = W.build_index(t, pan, k)
allk = W.union_of(keys)
g = torch.Generator(device=dev)
g.manual_seed(99)
dq, qk = W.presence_batch(allk, k, nq, g)
fp0 = t.footprint()
bits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
off = torch.zeros(nq + 1, dtype=torch.int64, device=dev)
need = torch.zeros(1, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
t.query_colors_dev(dq.data_ptr(), nq, bits.data_ptr(), off.data_ptr(), 0, 0, need.data_ptr(), st)  # the size first
torch.cuda.synchronize()
total = int(need.item())
ids = torch.zeros(total, dtype=torch.int32, device=dev)
call = lambda: t.query_colors_dev(dq.data_ptr(), nq, bits.data_ptr(), off.data_ptr(), ids.data_ptr(), total, need.data_ptr(), st)
call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    call()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
fp1 = t.footprint()
# ground truth on a slice: genome g is in the list of query i iff query i is a k-mer of genome g
ns = 1_000_000
o = off[: ns + 1].cpu().numpy()
idl = ids[: int(o[-1])].cpu().numpy()
got = np.zeros((ns, genomes), dtype=bool)
got[np.repeat(np.arange(ns), np.diff(o)), idl] = True
exp = np.zeros_like(got)
for gi, gkeys in enumerate(keys):
    pos = torch.searchsorted(gkeys, qk[:ns]).clamp(max=gkeys.numel() - 1)
    exp[:, gi] = (gkeys[pos] == qk[:ns]).cpu().numpy()
asc = bool(all((np.diff(idl[o[i]:o[i + 1]]) > 0).all() for i in range(0, 20000)))
print(json.dumps({"workload": f"{wl}: {genomes}-genome index, {nq} resident queries, id lists (bft_gpu_query_colors_dev)", "ms": round(ms, 3),
                  "M_kmers_per_s": round(nq / ms / 1e3, 1), "ids": total, "ids_per_kmer": round(total / nq, 2), "GB_written_per_s": round((total * 4 + nq * 8.125) / ms / 1e6, 1),
                  "lists_checked": ns, "lists_ok": bool((got == exp).all()), "ids_ascending": asc,
                  "image_bytes_per_kmer_before": round(sum(v for k_, v in fp0.items() if k_ not in ("insertion_log",)) / t.info()["kmers"], 2),
                  "image_bytes_per_kmer_after": round(sum(v for k_, v in fp1.items() if k_ not in ("insertion_log",)) / t.info()["kmers"], 2),
                  "sorted_table_bytes_after": fp1["kmer_table"]}))
