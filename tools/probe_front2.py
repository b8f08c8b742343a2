import os, sys, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BFT_GPU_LIB"] = os.path.join(os.getcwd(), "tools", "microbench", "libbft_gpu_f2prof.so")
import numpy as np
from bloomfiltertrie_amd import BFT, synth as S, _lib
k = 63
anc = S.random_genome(20000, 77)
gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 5000 + g), k)) for g in range(2000)]
lib = _lib.load() if hasattr(_lib, "load") else None
for rep in range(2):
    t = BFT(k)
    for g, km in enumerate(gk): t.insert_kmers(km, g)
    L = ctypes.CDLL(os.environ["BFT_GPU_LIB"])
    L.bft_front2_prof(None, 1)
    t.build()
    out = (ctypes.c_ulonglong * 12)()
    L.bft_front2_prof(out, 0)
    v = list(out)
    tot = sum(v[:10]) or 1
    names = ["staging", "hash keys", "radix passes", "order check", "flags (2 gathers)", "distinct to LDS", "rank", "lengths/scan", "output", "loop (list, bounds)"]
    print(json.dumps({n: round(x / tot, 3) for n, x in zip(names, v)}), tot)
    t.close()
