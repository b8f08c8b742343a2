import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bloomfiltertrie_amd import BFT, synth as S
k = 63
anc = S.random_genome(20000, 77)
gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.01, 5000 + g), k)) for g in range(2000)]
for rep in range(2):
    t = BFT(k)
    t.set_option("build_stages", 1)
    for g, km in enumerate(gk): t.insert_kmers(km, g)
    t.build()
    print([(n[:28], round(ms, 2)) for n, ms, _ in t.build_stages()][:4], t.info()["kmers"])
    t.close()
