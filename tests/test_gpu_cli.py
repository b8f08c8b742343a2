"""The C harness bft_gpu (csrc/bft_gpu_cli.c) end to end on the GPU box: `build` -> .bft -> `load -query_kmers` /
`-query_branching`, outputs compared byte for byte with the reference's format filled from the oracle."""
import os
import subprocess

import numpy as np
import pytest

from bloomfiltertrie_amd import _lib, synth as S

pytestmark = pytest.mark.gpu
CLI = os.path.join(_lib.CSRC, "bft_gpu")


def _write_ascii(path, kmers, k, extra_lines=()):
    lines = S.packed_to_ascii(kmers, k) + list(extra_lines)
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return lines


def test_build_load_query_csv(oracle_mod, tmp_path):
    k = 27
    anc = S.random_genome(20000, 4)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.02, 10 + g), k)) for g in range(3)]
    os.chdir(tmp_path)
    names = []
    for g, km in enumerate(gk):
        _write_ascii(tmp_path / f"genome{g}.kmers", km, k)
        names.append(f"genome{g}.kmers")
    (tmp_path / "list.txt").write_text("".join(str(tmp_path / n) + "\n" for n in names))
    out = subprocess.run([CLI, "build", str(k), "kmers", "list.txt", "out.bft"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    # the oracle (restated reference reader) loads the GPU-written file
    o = oracle_mod.OracleBFT.load_bft(str(tmp_path / "out.bft"))
    allk = S.distinct(np.concatenate(gk))
    assert o.stats()["kmers"] == len(allk)
    # queries: present, absent, and two invalid lines (all-0 rows, src/file_io.c:844-850)
    rng = np.random.default_rng(0)
    q = np.concatenate([allk[::5], S.snp_mutants(allk[::7], k, 3)])
    q = q[rng.permutation(len(q))]
    lines = _write_ascii(tmp_path / "queries.txt", q, k, extra_lines=["ACGTNNNNACGTACGTACGTACGTACG", "ACGT"])
    (tmp_path / "qlist.txt").write_text(str(tmp_path / "queries.txt") + "\n")
    out = subprocess.run([CLI, "load", "out.bft", "-query_kmers", "kmers", "qlist.txt", "-query_branching", "kmers", "qlist.txt"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    bits, off, ids = o.query_colors(q)
    pres = S.from_bits(bits, len(q))
    exp = ",".join(names) + "\n"
    for i in range(len(q)):
        have = set(ids[int(off[i]):int(off[i + 1])].tolist())
        exp += ",".join("1" if g in have else "0" for g in range(3)) + "\n"
    exp += "0,0,0\n0,0,0\n"
    exp = exp[:-1] + "\0"  # the final newline is overwritten by NUL (src/file_io.c:873-876)
    got = (tmp_path / "queries.csv").read_bytes().decode()
    assert got == exp
    assert f"Nb k-mers present = {int(pres.sum())}" in out.stdout
    _, _, nbr = o.query_branching(q)
    assert f"Nb branching k-mers = {nbr}" in out.stdout
    # the same command with the batches sharded over a device group (BFT_GPU_DEVICES; two slots on the one GPU of this box): same bytes
    os.rename(tmp_path / "queries.csv", tmp_path / "queries_one.csv")
    out2 = subprocess.run([CLI, "load", "out.bft", "-query_kmers", "kmers", "qlist.txt", "-query_branching", "kmers", "qlist.txt"],
                          capture_output=True, text=True, env=dict(os.environ, BFT_GPU_DEVICES="0,0"))
    assert out2.returncode == 0, out2.stderr
    assert (tmp_path / "queries.csv").read_bytes() == (tmp_path / "queries_one.csv").read_bytes()
    assert f"Nb k-mers present = {int(pres.sum())}" in out2.stdout and f"Nb branching k-mers = {nbr}" in out2.stdout


def test_kmers_comp_input_and_bad_k(tmp_path):
    k = 18
    km = S.distinct(S.kmers_of(S.random_genome(5000, 1), k))
    os.chdir(tmp_path)
    with open("g.kmers_comp", "wb") as f:
        f.write(f"{k}\n{len(km)}\n".encode())
        f.write(km.tobytes())
    (tmp_path / "list.txt").write_text(str(tmp_path / "g.kmers_comp") + "\n")
    assert subprocess.run([CLI, "build", str(k), "kmers_comp", "list.txt", "o.bft"], capture_output=True).returncode == 0
    with open("q.kmers_comp", "wb") as f:
        f.write(f"{k}\n{len(km)}\n".encode())
        f.write(km[:1000].tobytes())
    (tmp_path / "ql.txt").write_text(str(tmp_path / "q.kmers_comp") + "\n")
    out = subprocess.run([CLI, "load", "o.bft", "-query_kmers", "kmers_comp", "ql.txt"], capture_output=True, text=True)
    assert out.returncode == 0 and "Nb k-mers present = 1000" in out.stdout
    bad = subprocess.run([CLI, "build", "31", "kmers", "list.txt", "x.bft"], capture_output=True, text=True)
    assert bad.returncode != 0 and "multiple of 9" in bad.stderr


def test_query_sequences_csv(oracle_mod, tmp_path):
    k, ngen = 27, 4
    anc = S.random_genome(6000, 14)
    strs = ["".join("ACGT"[c] for c in S.mutate(anc, 0.03, 80 + g)) for g in range(ngen)]
    os.chdir(tmp_path)
    names = []
    o = oracle_mod.OracleBFT(k)
    for g, s in enumerate(strs):
        km = S.distinct(S.kmers_of(S._CODE[np.frombuffer(s.encode(), dtype=np.uint8)], k))
        _write_ascii(tmp_path / f"g{g}.kmers", km, k)
        names.append(f"g{g}.kmers")
        o.insert_kmers(km, g)
    (tmp_path / "list.txt").write_text("".join(str(tmp_path / n) + "\n" for n in names))
    assert subprocess.run([CLI, "build", str(k), "kmers", "list.txt", "o.bft"], capture_output=True).returncode == 0
    rng = np.random.default_rng(2)
    reads = [strs[int(rng.integers(0, ngen))][a:a + int(rng.integers(30, 250))] for a in rng.integers(0, 5000, 60)]
    reads += ["ACGTACGT", "N" * 40]
    (tmp_path / "reads.txt").write_text("\n".join(reads) + "\n")
    (tmp_path / "rl.txt").write_text(str(tmp_path / "reads.txt") + "\n")
    out = subprocess.run([CLI, "load", "o.bft", "-query_sequences", "0.6", "non_canonical", "rl.txt"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    exp = ",".join(names) + "\n"
    for r in reads:
        have = set(o.query_sequence(r, 0.6, False, ngen))
        exp += ",".join("1" if g in have else "0" for g in range(ngen)) + "\n"
    exp = exp[:-1] + "\0"
    assert (tmp_path / "reads.csv").read_bytes().decode() == exp


def test_add_genomes(oracle_mod, tmp_path):
    """bft load X -add_genomes kmers list out (src/main.c:217-246): the extended index equals a one-shot build."""
    k = 18
    anc = S.random_genome(8000, 31)
    gk = [S.distinct(S.kmers_of(S.mutate(anc, 0.03, 5 + g), k)) for g in range(4)]
    os.chdir(tmp_path)
    for g, km in enumerate(gk):
        _write_ascii(tmp_path / f"g{g}.kmers", km, k)
    (tmp_path / "l01.txt").write_text("".join(str(tmp_path / f"g{g}.kmers") + "\n" for g in (0, 1)))
    (tmp_path / "l23.txt").write_text("".join(str(tmp_path / f"g{g}.kmers") + "\n" for g in (2, 3)))
    assert subprocess.run([CLI, "build", str(k), "kmers", "l01.txt", "a.bft"], capture_output=True).returncode == 0
    out = subprocess.run([CLI, "load", "a.bft", "-add_genomes", "kmers", "l23.txt", "b.bft"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    o = oracle_mod.OracleBFT.load_bft(str(tmp_path / "b.bft"))
    ref = oracle_mod.OracleBFT(k)
    for g, km in enumerate(gk):
        ref.insert_kmers(km, g)
    allk = S.distinct(np.concatenate(gk))
    q = np.concatenate([allk, S.snp_mutants(allk[::3], k, 1)])
    assert all((a == b).all() for a, b in zip(o.query_colors(q), ref.query_colors(q)))
    assert o.nb_genomes_loaded() == 4


def test_extract_kmers(tmp_path):
    k = 27
    km = S.distinct(S.kmers_of(S.random_genome(4000, 2), k))
    os.chdir(tmp_path)
    _write_ascii(tmp_path / "g.kmers", km, k)
    (tmp_path / "l.txt").write_text(str(tmp_path / "g.kmers") + "\n")
    assert subprocess.run([CLI, "build", str(k), "kmers", "l.txt", "o.bft"], capture_output=True).returncode == 0
    out = subprocess.run([CLI, "load", "o.bft", "-extract_kmers", "kmers", "x.txt", "-extract_kmers", "kmers_comp", "x.bin"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = [l for l in (tmp_path / "x.txt").read_text().split("\n") if l]
    assert sorted(got) == sorted(S.packed_to_ascii(km, k))
    raw = (tmp_path / "x.bin").read_bytes()
    head = f"{k}\n{len(km)}\n".encode()
    assert raw.startswith(head)
    packed = np.frombuffer(raw[len(head):], dtype=np.uint8).reshape(-1, 7)
    assert sorted(map(bytes, packed)) == sorted(map(bytes, km))
