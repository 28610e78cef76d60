"""Build libsph_hip.so (hipcc, gfx950) in-tree.  `python -m gpufluidsimulator_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsph_hip.so")
HIP_SOURCES = ["sph_capi.hip", "sph_sort.hip", "sph_pairs.hip", "sph_halo.hip", "sph_slab.hip", "sph_compat.hip"]
CXX_SOURCES = ["particleSystem.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP pass packs neighbouring fp32 adds/multiplies into v_pk_*_f32, which on
# gfx950 run at the scalar rate and cost extra v_mov shuffles (measured on k_force: 4.34 -> 3.45 ms at C3)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-Wno-unused-value",
         "-fno-slp-vectorize",
         "-I" + os.path.join(ROOT, "include")]


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP/C++ source of the library for gfx950 and link libsph_hip.so.
    hipcc cross-compiles without a GPU, so this also runs in the CPU-only container."""
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    srcs = [s for s in HIP_SOURCES + CXX_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            if s in CXX_SOURCES:   # plain host C++ (no HIP headers); ICs must match ic.py bit for bit
                jobs.append([HIPCC, "-O2", "-std=c++17", "-fPIC", "-Wall", "-ffp-contract=off", "-x", "c++",
                             "-c", src, "-o", obj])
            else:
                jobs.append([HIPCC] + FLAGS + ["-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs + ["-ldl"])
    # headless driver with the reference's command line (SPH/particles.cpp)
    exe, main_src = os.path.join(HERE, "sph_headless"), os.path.join(CSRC, "sph_headless.cpp")
    if os.path.exists(main_src) and (force or _stale(exe, [main_src, LIB] + headers)):
        run([HIPCC, "-O2", "-std=c++17", "-x", "c++", main_src, "-x", "none", "-o", exe, "-L" + HERE, "-lsph_hip",
             "-Wl,-rpath,$ORIGIN"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-q" not in sys.argv))
