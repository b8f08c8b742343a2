"""world_size-2 gloo test of the N>1 path: contiguous query shards per rank, replicated trie, one
all_gather of the presence bitmaps.  On CPU the per-rank "query" is the oracle (the checker), the sharding and
gather code is the product's (bloomfiltertrie_amd/dist.py)."""
import os
import socket

import numpy as np
import pytest

from bloomfiltertrie_amd import synth as S
from bloomfiltertrie_amd.dist import shard_bounds


def test_shard_bounds_cover_and_align():
    for n in (0, 1, 63, 64, 65, 1000, 12345, 10 ** 6 + 7):
        for ws in (1, 2, 3, 4, 8):
            seen = 0
            for r in range(ws):
                a, b, per = shard_bounds(n, ws, r)
                assert (a % 64 == 0 or a == n) and per % 64 == 0 and a <= b <= n
                assert a == min(n, r * per)
                seen += b - a
            assert seen == n


def _worker(rank, world, port, k, km, q, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from bloomfiltertrie_amd.dist import query_presence_sharded

    class FakeBFT:  # stands in for the GPU handle: same query_presence contract, answered by the oracle
        device = 0

        def __init__(self):
            self.o = O.OracleBFT(k)
            self.o.insert_kmers(km, 0)

        def query_presence(self, kmers):
            return self.o.query_presence(np.ascontiguousarray(kmers))

    bits = query_presence_sharded(FakeBFT(), q)
    ret[rank] = bits.tobytes()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq", [1000, 12345])
def test_sharded_query_gloo_world2(oracle_mod, nq):
    import torch.multiprocessing as mp
    k = 27
    km = S.distinct(S.kmers_of(S.random_genome(20000, 1), k))
    rng = np.random.default_rng(0)
    q = np.concatenate([km[: nq // 2], S.pack_codes(rng.integers(0, 4, (nq - nq // 2, k), dtype=np.uint8))])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, k, km, q, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    exp = S.to_bits(S.member(q, km)).tobytes()
    assert ret[0] == exp and ret[1] == exp


def _pipeline_worker(rank, world, port, nq, steps, ret):
    """bench.py's step / drain loop (bloomfiltertrie_amd.dist.GatherPipeline) with a fake handle on gloo: the query of
    step i writes a pattern that depends on (rank, i); after every step the gathered buffer the pipeline reports must hold
    exactly the patterns of that step from every rank, although the gather of step i overlaps the query of step i+1."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bloomfiltertrie_amd.dist import GatherPipeline
    nbytes = ((nq + 63) // 64) * 8
    calls = []

    def fake_query(buf):  # stands in for bft.query_presence_dev on the rank's resident shard
        i = len(calls)
        calls.append(buf.data_ptr())
        buf.copy_(torch.full((nbytes,), (17 * rank + 3 * i + 1) % 251, dtype=torch.uint8))

    pipe = GatherPipeline(fake_query, nbytes, world, torch.device("cpu"), use_dist=True)
    ok = pipe.nbuf == 2
    for i in range(steps):
        pipe.step()
        if i % 2 == 1 or i == steps - 1:
            pipe.drain()
            local, gathered = pipe.last()
            for r in range(world):
                exp = (17 * r + 3 * i + 1) % 251
                ok = ok and bool((gathered[r * nbytes:(r + 1) * nbytes] == exp).all())
            ok = ok and bool((local == (17 * rank + 3 * i + 1) % 251).all())
    pipe.drain()
    ok = ok and len(calls) == steps and len(set(calls)) == 2 and calls[0] != calls[1] and calls[0] == calls[2]  # two buffers, alternating
    ret[rank] = ok
    dist.destroy_process_group()


def test_bench_gather_pipeline_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, 100_003, 5, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] is True and ret[1] is True


def test_bench_gather_pipeline_single_rank_has_no_collective():
    import torch
    from bloomfiltertrie_amd.dist import GatherPipeline
    seen = []
    pipe = GatherPipeline(lambda buf: seen.append(buf.data_ptr()) or buf.fill_(5), 64, 1, torch.device("cpu"), use_dist=False)
    for _ in range(3):
        pipe.step()
    pipe.drain()
    local, gathered = pipe.last()
    assert gathered is None and pipe.nbuf == 1 and len(set(seen)) == 1 and bool((local == 5).all())


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` outside torchrun must start N rank processes itself: checked without a GPU by
    pointing the spawner at a stub interpreter that records the rank environment it was given."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json; sys.path.insert(0, %r); sys.argv=['bench.py','--gpus','3','--steps','1']; import bench, os\n"
            "class P:\n"
            "    def __init__(self, cmd, env=None, stdout=None):\n"
            "        self.rec = {k: env[k] for k in ('RANK','LOCAL_RANK','WORLD_SIZE','MASTER_ADDR','MASTER_PORT')}; self.rec['cmd'] = cmd[1:]; recs.append(self.rec)\n"
            "    def wait(self): return 0\n"
            "recs = []; bench.subprocess.Popen = P; rc = bench.spawn_ranks(bench.parse()); print(json.dumps({'rc': rc, 'recs': recs}))") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().split("\n")[-1])
    assert d["rc"] == 0 and [x["RANK"] for x in d["recs"]] == ["0", "1", "2"] and {x["WORLD_SIZE"] for x in d["recs"]} == {"3"}
    assert {x["MASTER_ADDR"] for x in d["recs"]} == {"127.0.0.1"} and len({x["MASTER_PORT"] for x in d["recs"]}) == 1
    assert all(x["cmd"][0].endswith("bench.py") and "--gpus" in x["cmd"] for x in d["recs"])


def _rows_worker(rank, world, port, k, genomes, q, ret):
    """colour rows and branching through the sharded calls on gloo: the per-rank answers come from the oracle (the checker), the
    slicing and the two gathers -- bitmaps and fixed-width rows -- are the product's."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from bloomfiltertrie_amd.dist import query_branching_sharded, query_color_rows_sharded

    class FakeBFT:
        device = 0

        def __init__(self):
            self.o = O.OracleBFT(k)
            for g, km in enumerate(genomes):
                self.o.insert_kmers(km, g)

        def query_color_rows(self, kmers):
            bits, off, ids = self.o.query_colors(np.ascontiguousarray(kmers))
            rows = np.zeros((len(kmers), (len(genomes) + 7) // 8), np.uint8)
            for i in range(len(kmers)):
                for g in ids[int(off[i]):int(off[i + 1])]:
                    rows[i, g >> 3] |= 1 << (g & 7)
            return bits, rows

        def query_branching(self, kmers, with_counts=False):
            bits, counts, _ = self.o.query_branching(np.ascontiguousarray(kmers))
            return (bits, counts) if with_counts else bits

    f = FakeBFT()
    bits, rows = query_color_rows_sharded(f, q, len(genomes))
    bbits, counts = query_branching_sharded(f, q)
    ret[rank] = (bits.tobytes(), rows.tobytes(), rows.shape, bbits.tobytes(), counts.tobytes())
    dist.destroy_process_group()


@pytest.mark.parametrize("nq", [700, 4099])
def test_sharded_colour_rows_and_branching_gloo_world2(oracle_mod, nq):
    """colour and branching queries shard like presence queries -- contiguous 64-aligned slices, the fixed-width rows
    gathered beside the bitmaps.  Both ranks end with the answers of the whole batch: genome bitmaps == the inserting genomes,
    branching counts == the oracle's on the unsharded batch."""
    import torch.multiprocessing as mp
    k = 18
    anc = S.random_genome(6000, 3)
    genomes = [S.distinct(S.kmers_of(S.mutate(anc, 0.03, 40 + g), k)) for g in range(11)]
    allk = S.distinct(np.concatenate(genomes))
    rng = np.random.default_rng(1)
    q = np.concatenate([allk[rng.choice(len(allk), nq // 2)], S.pack_codes(rng.integers(0, 4, (nq - nq // 2, k), dtype=np.uint8))])
    q = np.ascontiguousarray(q[rng.permutation(len(q))])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_rows_worker, args=(r, 2, port, k, genomes, q, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    exp_rows = np.zeros((nq, 2), np.uint8)
    for g, km in enumerate(genomes):
        exp_rows[S.member(q, km), g >> 3] |= 1 << (g & 7)
    o = oracle_mod.OracleBFT(k)
    for g, km in enumerate(genomes):
        o.insert_kmers(km, g)
    ebits, ecounts, _ = o.query_branching(q)
    for r in range(2):
        bits, rows, shape, bbits, counts = ret[r]
        assert shape == (nq, 2)
        assert bits == S.to_bits(S.member(q, allk)).tobytes()
        assert rows == exp_rows.tobytes()
        assert bbits == ebits.tobytes() and counts == ecounts.tobytes()
