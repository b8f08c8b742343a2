#!/usr/bin/env python3
"""Mutated .bft files through the product's reader (csrc/bft_file.cpp, the host-only test library -- no GPU): truncations, flipped bytes, extreme
values in 2- and 4-byte fields.  The reader must reject or accept every one of them without crashing; a worker process per batch, so that a crash
is a result (the batch's seed) rather than the end of the run.
usage: fuzz_bft_reader.py [batches] [mutations per batch] [first seed]      worker: fuzz_bft_reader.py --worker <seed> <n> <dir>"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from bloomfiltertrie_amd import _lib, synth as S  # noqa: E402


def hostlib():
    subprocess.check_call(["make", "-C", _lib.CSRC, "libbft_hosttest.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(_lib.CSRC, "libbft_hosttest.so"))
    lib.bft_hosttest_build.restype = C.c_void_p
    lib.bft_hosttest_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int]
    lib.bft_hosttest_free.argtypes = [C.c_void_p]
    lib.bft_hosttest_write_bft.argtypes = [C.c_void_p, C.c_char_p]
    lib.bft_hosttest_read_bft.restype = C.c_void_p
    lib.bft_hosttest_read_bft.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    lib.bft_hosttest_read_free.argtypes = [C.c_void_p]
    return lib


def seeds_files(lib, d):
    """a few valid files: one and several levels, UC-only and CC roots"""
    out = []
    for i, (k, levels, n) in enumerate([(9, 1, 300), (18, 1, 5000), (27, 2, 3000), (36, 3, 2000), (63, 3, 800)]):
        km = S.low_entropy_kmers(n, k, 10, seed=i + 1, levels=levels) if levels > 1 else S.distinct(S.kmers_of(S.random_genome(n, i + 3), k))
        km = np.ascontiguousarray(km)
        h = lib.bft_hosttest_build(km.ctypes.data, len(km), k, 3, 0)
        assert h
        p = os.path.join(d, f"seed{i}.bft")
        lib.bft_hosttest_write_bft(h, p.encode())
        lib.bft_hosttest_free(h)
        out.append(p)
    return out


def worker(seed, n, d):
    lib = hostlib()
    files = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.startswith("seed")]
    rng = np.random.default_rng(seed)
    p = os.path.join(d, f"mut{seed}.bft")
    accepted = 0
    for it in range(n):
        b = bytearray(open(files[int(rng.integers(0, len(files)))], "rb").read())
        kind = int(rng.integers(0, 4))
        if kind == 0:
            b = b[: int(rng.integers(0, len(b)))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 2:
            at = int(rng.integers(0, max(1, len(b) - 4)))
            b[at:at + 4] = int(rng.choice([0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0x00FFFFFF, 0x01000000])).to_bytes(4, "little")
        else:
            at = int(rng.integers(0, max(1, len(b) - 2)))
            b[at:at + 2] = int(rng.choice([0xFFFF, 0x7FFF, 0x8000, 0x00FF, 0xFFFE])).to_bytes(2, "little")
        open(p, "wb").write(bytes(b))
        print(f"{seed} {it} {kind}", flush=True)  # (the last line before a crash names the case)
        k, g, m = C.c_int(), C.c_int(), C.c_uint64()
        h = lib.bft_hosttest_read_bft(p.encode(), C.byref(k), C.byref(g), C.byref(m))
        if h:
            accepted += 1
            lib.bft_hosttest_read_free(h)
    print(f"done {seed} accepted {accepted} of {n}", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        return
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lib = hostlib()
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        seeds_files(lib, d)
        bad = 0
        for s in range(first, first + batches):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(s), str(per), d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
            lines = r.stdout.decode(errors="replace").strip().splitlines()
            if r.returncode != 0:
                bad += 1
                print(f"CRASH rc={r.returncode} last case: {lines[-1] if lines else '?'}", flush=True)
            else:
                print(lines[-1], flush=True)
        print("fuzz", "FAILED" if bad else "OK", f"({batches} batches of {per})")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
