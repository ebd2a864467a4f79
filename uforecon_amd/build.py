"""Build recipe for libufr.so (hipcc, gfx950 only, in-tree so the .so travels with the snapshot).

``python -m uforecon_amd.build`` or ``build_library()``.  Objects are rebuilt only when a source
or header is newer than them.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OUT_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(OUT_DIR, "libufr.so")
ARCH = "gfx950"

SOURCES = ["ufr_api.hip", "prep.hip", "sampler.hip", "gather.hip", "view_transformer.hip",
           "ray_transformer.hip", "composite.hip"]
CXXFLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
            f"-I{INCLUDE}", f"-I{CSRC}"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libufr.so cannot be built on this host")


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(INCLUDE, "ufr.h"))
    return max(os.path.getmtime(h) for h in hs)


def build_library(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_t = _newest_header()
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_t):
            jobs.append([hipcc, *CXXFLAGS, *extra_flags, "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB_PATH):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH])
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
