#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel in libbft_gpu.so, read from the code objects' metadata notes
(clang offload bundles inside the .so -> gfx950 ELF -> llvm-readelf --notes).  No GPU needed.
usage: kernel_resources.py [substring of the kernel name] [--lib path]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    pos = 0
    while True:
        at = blob.find(MAGIC, pos)
        if at < 0:
            return
        n = struct.unpack_from("<Q", blob, at + len(MAGIC))[0]
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[at + off:at + off + size]
        pos = at + len(MAGIC)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = os.path.join(ROOT, "bloomfiltertrie_amd", "csrc", "libbft_gpu.so")
    if "--lib" in sys.argv:
        lib = sys.argv[sys.argv.index("--lib") + 1]
        args = [a for a in args if a != lib]
    want = args[0] if args else ""
    blob = open(lib, "rb").read()
    rows = []
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for m in re.finditer(r"- \.agpr_count:.*?(?=\n\s+- \.agpr_count:|\namdhsa\.target|\Z)", notes, re.S):
            txt = m.group(0)
            get = lambda key: (re.search(r"\." + key + r":\s+(\S+)", txt) or [None, "?"])[1]
            rows.append((get("name"), get("vgpr_count"), get("sgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"),
                         get("private_segment_fixed_size"), get("group_segment_fixed_size"), get("max_flat_workgroup_size")))
    names = subprocess.run([CXXFILT], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'vgpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>6} {'maxwg':>5}  kernel")
    for r, nm in sorted(zip(rows, names), key=lambda x: x[1]):
        if want in nm:
            print(f"{r[1]:>5} {r[2]:>5} {r[3]:>6} {r[4]:>6} {r[5]:>7} {r[6]:>6} {r[7]:>5}  {nm[:150]}")


if __name__ == "__main__":
    main()
