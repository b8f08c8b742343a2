#!/usr/bin/env python3
"""Variants of the container walk (k_query*) on the config-4 share: the 100-genome index, 1.25x10^8 presence queries, k-mer hash off.
For every variant: mean launch time over `reps` launches (HIP events of the library) and whether the bitmap equals the k-mer hash's.
usage: probe_walk.py [out.jsonl] [--cfg2] [--k K] [variant ...]     a variant = comma-separated option=value pairs
Default variants: probe rows x root quartiles x claims x round size x residency."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bloomfiltertrie_amd import BFT, workloads as W  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
out_path = args[0] if args else os.path.join(ROOT, "gpurun_out", "probe_walk.jsonl")
k = 27
for i, f in enumerate(sys.argv):
    if f == "--k":
        k = int(sys.argv[i + 1])
        args = [a for a in args if a != sys.argv[i + 1]]
variants = args[1:]
genomes = 10 if "--cfg2" in flags else 100
nq = 100_000_000 if "--cfg2" in flags else 125_000_000
reps = 5
dev = torch.device("cuda", 0)
pan = W.PanGenome(genomes, 2_000_000, 0.01, 4242, dev)
t = BFT(k)
keys, _ = W.build_index(t, pan, k)
allk = W.union_of(keys)
del keys
g = torch.Generator(device=dev)
g.manual_seed(99)
dq, qk = W.presence_batch(allk, k, nq, g)
stream = torch.cuda.current_stream().cuda_stream
ref = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
t.query_presence_dev(dq.data_ptr(), nq, ref.data_ptr(), stream)
torch.cuda.synchronize()
truth_ok = bool((W.bits_to_bool(ref, nq) == W.member(allk, qk)).all())


def timed(bits):
    t.query_presence_dev(dq.data_ptr(), nq, bits.data_ptr(), stream)
    torch.cuda.synchronize()
    t.kernel_time(reset=True)
    for _ in range(reps):
        t.query_presence_dev(dq.data_ptr(), nq, bits.data_ptr(), stream)
    torch.cuda.synchronize()
    ms, n = t.kernel_time(reset=True)
    return ms / max(1, n)


if not variants:
    variants = ["kmer_hash=1", "walk_hash=1", "kmer_hash=0", "kmer_hash_load=50", "kmer_hash_load=70", "kmer_hash_load=80", "compact_table=1", "walk_hash=1,query_wgs_per_cu=1",
                "walk_hash=1,query_wgs_per_cu=2", "query_dynamic=0"]
if False:
    variants = [
        "query_dynamic=0,root_quartiles=0,query_probe=8",
        "query_dynamic=1,root_quartiles=0,query_probe=8",
        "query_dynamic=1,root_quartiles=0,query_probe=16",
        "query_dynamic=1,root_quartiles=0,query_probe=4",
        "query_dynamic=1,root_quartiles=1,query_probe=4",
        "query_dynamic=1,root_quartiles=1,query_probe=8",
        "query_dynamic=1,root_quartiles=1,query_probe=16",
        "query_dynamic=0,root_quartiles=1,query_probe=8",
        "query_dynamic=1,root_quartiles=1,query_probe=8,walk_chunk=2",
        "query_dynamic=1,root_quartiles=1,query_probe=8,walk_chunk=1",
        "query_dynamic=1,root_quartiles=1,query_probe=8,query_wgs_per_cu=1",
        "query_dynamic=1,root_quartiles=1,query_probe=8,query_wgs_per_cu=2",
        "query_dynamic=1,root_quartiles=1,query_probe=16,query_wgs_per_cu=1",
        "query_dynamic=1,root_quartiles=1,query_probe=16,query_wgs_per_cu=2",
        "query_dynamic=1,root_quartiles=1,query_probe=8,node_hash=2",
    ]
kh_ms = timed(torch.zeros_like(ref))
rows = [{"variant": "kmer_hash (default path)", "ms": round(kh_ms, 4), "G_kmers_per_s": round(nq / kh_ms / 1e6, 2), "truth_ok": truth_ok, "k": k, "genomes": genomes}]
print(rows[-1], flush=True)
defaults = {"kmer_hash_load": 60, "kmer_hash": 1, "walk_hash": 0, "query_dynamic": 1, "root_quartiles": 1, "query_probe": 0, "query_wgs_per_cu": 0, "query_grid_mult": 1,
            "root_direct": 3, "node_hash": 1, "compact_table": 0}
for v in variants:
    opts = dict(defaults)
    opts.update({a.split("=")[0]: int(a.split("=")[1]) for a in v.split(",")})
    for name, val in opts.items():
        t.set_option(name, val)
    bits = torch.zeros_like(ref)
    ms = timed(bits)
    fp = t.footprint()
    rows.append({"variant": v, "ms": round(ms, 4), "G_kmers_per_s": round(nq / ms / 1e6, 2), "same_bits": bool(torch.equal(bits, ref)),
                 "image_B_per_kmer": round(t.info()["image_bytes"] / t.info()["kmers"], 2), "kh_B_per_kmer": round(fp["kmer_hash"] / t.info()["kmers"], 2),
                 "kh_lines": t.build_time()["kmer_hash_lines"]})
    print(rows[-1], flush=True)
    del bits
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "a") as f:
    for r in rows:
        f.write(json.dumps(r) + "\n")
