#!/usr/bin/env python3
"""`make clean && make -jN all` of bloomfiltertrie_amd/csrc (hipcc --offload-arch=gfx950: cross-compiles without a GPU) and of oracle/, timed:
the evidence that the tree builds from source (the .so / .o files are git-ignored and travel to the GPU box prebuilt).
usage: clean_build_time.py [out.json] [jobs]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05", "make_clean_build.json")
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 8)
csrc = os.path.join(ROOT, "bloomfiltertrie_amd", "csrc")
res = {"command": f"make -C bloomfiltertrie_amd/csrc clean && make -j{jobs} -C bloomfiltertrie_amd/csrc all && make -C oracle clean all", "jobs": jobs, "cpus": os.cpu_count()}
subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
t0 = time.time()
r = subprocess.run(["make", f"-j{jobs}", "-C", csrc, "all"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
res["csrc_seconds"] = round(time.time() - t0, 1)
res["csrc_rc"] = r.returncode
res["csrc_warnings"] = r.stdout.decode(errors="replace").count("warning:")
t0 = time.time()
subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "clean"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
r2 = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
res["oracle_seconds"] = round(time.time() - t0, 1)
res["oracle_rc"] = r2.returncode
res["built"] = sorted(f for f in os.listdir(csrc) if f.endswith(".so") or f == "bft_gpu")
res["hipcc"] = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode(errors="replace").splitlines()[0]
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
