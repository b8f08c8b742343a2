"""Bucketed batches against the derived tables: config 2, 10^8 queries, query_bucket_bits = 8, every combination of the
hashed groups / root tables / node prefix hash.  BFT_GPU_LIB selects another build of the library (bisecting)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from bloomfiltertrie_amd import BFT, workloads as W
from bloomfiltertrie_amd._lib import BFTError
dev = torch.device("cuda", 0)
k, nq = 27, 100_000_000
pan = W.PanGenome(10, 2_000_000, 0.01, 4242, dev)
t = BFT(k); keys, _ = W.build_index(t, pan, k); allk = W.union_of(keys)
g = torch.Generator(device=dev); g.manual_seed(99)
dq, qk = W.presence_batch(allk, k, nq, g)
dbits = torch.zeros(((nq + 63) // 64) * 8, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def opt(name, v):
    try: t.set_option(name, v); return True
    except BFTError: return False
def run(tag):
    t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): t.query_presence_dev(dq.data_ptr(), nq, dbits.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("BFT_GPU_LIB", "current"), tag, round(e0.elapsed_time(e1) / 3, 3), flush=True)
quick = "--quick" in sys.argv
for bits in (0, 8):
    opt("query_bucket_bits", bits)
    for nh in ((1,) if quick else (1, 0)):
        opt("node_hash", nh)
        for gh in ((1,) if quick else (1, 0)):
            for rd in ((2,) if quick else (2, 1, 0)):
                opt("group_hash", gh); opt("root_direct", rd)
                run(f"bits={bits} nh={nh} gh={gh} rd={rd}")
