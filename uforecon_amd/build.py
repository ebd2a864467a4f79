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
           "ray_transformer.hip", "composite.hip", "render_loss.hip", "view_dgrad.hip", "wgrad_stream.hip", "ray_dgrad.hip", "presim_bwd.hip", "gather_bwd.hip", "frustum.hip", "tsdf.hip", "dcn.hip", "fmt.hip", "conv3d.hip", "conv3d_planes.hip", "conv3d_wgrad_planes.hip", "conv2d.hip"]
# -ffp-contract=on: fuse a*b+c only inside one expression.  hipcc's default (fast) also fuses across statements,
# and did so differently in the two unrolled copies of the per-tile code of the view transformer: a point's result
# then depended on which column tile it landed in (1 ulp), which breaks "rays are independent -> chunking and the
# fine pass's reuse of coarse evaluations are invisible" (tests/test_gpu_parity.py checks both bit for bit).
# -fno-slp-vectorize: packed-f32 VALU (v_pk_add/fma_f32) issued beside MFMAs costs more than the two scalar
# instructions it replaces (MI355X_MICROARCH.md; measured 2 % on both transformer kernels)
CXXFLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
            "-fno-slp-vectorize", "-ffp-contract=on", f"-I{INCLUDE}", f"-I{CSRC}"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libufr.so cannot be built on this host")


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(INCLUDE, "ufr.h"))
    return max(os.path.getmtime(h) for h in hs)


def build_library(force: bool = False, verbose: bool = True, extra_flags=(), variant: str = "") -> str:
    """variant != "" builds lib/libufr_<variant>.so from separate objects (development A/B builds)."""
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = _hipcc()
    hdr_t = _newest_header()
    jobs = []
    objs = []
    suffix = f"_{variant}" if variant else ""
    lib_path = os.path.join(OUT_DIR, f"libufr{suffix}.so")
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OUT_DIR, src.replace(".hip", f"{suffix}.o"))
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
    stale = not os.path.exists(lib_path) or any(os.path.getmtime(o) > os.path.getmtime(lib_path) for o in objs)
    if jobs or stale:   # an interrupted build can leave fresh objects beside an old library
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", lib_path])
    return lib_path


if __name__ == "__main__":
    # python -m uforecon_amd.build [--force] [--variant NAME -DFLAG ...]
    argv = sys.argv[1:]
    variant = argv[argv.index("--variant") + 1] if "--variant" in argv else ""
    flags = [a for a in argv if a.startswith(("-D", "-m", "-f"))]
    print(build_library(force="--force" in argv, verbose=False, extra_flags=flags, variant=variant))
