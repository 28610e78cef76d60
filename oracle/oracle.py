"""ctypes loader for the CPU oracle (oracle/sph_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

PARTICLE_DTYPE = np.dtype([
    ("index", np.uint32), ("position", np.float32, 3), ("velocity", np.float32, 3),
    ("delta_velocity", np.float32, 3), ("force_press", np.float32, 3), ("force_visc", np.float32, 3),
    ("mass", np.float32), ("density", np.float32), ("pressure", np.float32), ("radius", np.float32),
    ("collision_count", np.int32), ("zindex", np.uint32)])
assert PARTICLE_DTYPE.itemsize == 88
GRID_DTYPE = np.dtype([("nParticles", np.uint32), ("start", np.uint32)])

CELL_MORTON, CELL_LINEAR = 0, 1


class _Sys(C.Structure):
    _fields_ = [("n", C.c_uint32), ("p", C.c_void_p), ("grid", C.c_uint32 * 3), ("cell_mode", C.c_uint32),
                ("b_size", C.c_uint32), ("B", C.c_void_p), ("bprime_size", C.c_uint32), ("Bprime", C.c_void_p),
                ("box_min", C.c_float * 3), ("box_max", C.c_float * 3), ("box_dims", C.c_float * 3),
                ("hpos", C.c_void_p), ("occ", C.c_void_p), ("n_occ", C.c_uint32), ("tmp", C.c_void_p),
                ("keybuf", C.c_void_p)]


def build(fast: bool = False) -> str:
    """Compile the oracle (a few seconds) if the .so is missing or stale."""
    name = "liboracle_fast.so" if fast else "liboracle.so"
    so = os.path.join(HERE, name)
    src = [os.path.join(HERE, f) for f in ("sph_oracle.c", "sph_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", HERE, name])
    return so


def cpu_info():
    """(model name, logical cpus available to this process) from /proc/cpuinfo and the affinity mask."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return model, cores


def build_native() -> str:
    """The speed build (-O3 -march=native) compiled ON THE MACHINE THAT RUNS IT, one file per CPU model
    under oracle/_native/: a -march=native object built in the build container must not be run on the
    GPU box's (different) host CPU.  bench.py's cpu_baseline "port" leg uses this."""
    import hashlib
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    tag = hashlib.sha1((cpu_info()[0] + flags).encode()).hexdigest()[:10]
    out_dir = os.path.join(HERE, "_native")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, f"liboracle_fast_{tag}.so")
    src = [os.path.join(HERE, f) for f in ("sph_oracle.c", "sph_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-std=c99", "-shared", "-o", so,
                               src[0], "-lm"])
    return so


_libs = {}


def lib(fast=False):
    """fast: False = the bit-exact build, True = liboracle_fast.so, "native" = build_native()."""
    if fast not in _libs:
        L = C.CDLL(build_native() if fast == "native" else build(fast))
        L.orc_create.restype = C.POINTER(_Sys)
        L.orc_create.argtypes = [C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_uint32]
        for fn in ("orc_destroy", "orc_map_zindex", "orc_sort", "orc_construct_bgrid", "orc_construct_grid_array",
                   "orc_compute_densities", "orc_compute_forces", "orc_particle_collisions",
                   "orc_compute_densities_n2"):
            getattr(L, fn).restype = None
            getattr(L, fn).argtypes = [C.POINTER(_Sys)]
        L.orc_integrate.restype = None
        L.orc_integrate.argtypes = [C.POINTER(_Sys), C.c_float]
        L.orc_step.restype = None
        L.orc_step.argtypes = [C.POINTER(_Sys), C.c_float]
        L.orc_load.restype = None
        L.orc_load.argtypes = [C.POINTER(_Sys), C.c_void_p, C.c_void_p]
        L.orc_apply_order.restype = None
        L.orc_apply_order.argtypes = [C.POINTER(_Sys), C.c_void_p]
        L.orc_coord2zindex.restype = C.c_uint32
        L.orc_coord2zindex.argtypes = [C.c_uint32] * 3
        L.orc_zindex2coord.restype = None
        L.orc_zindex2coord.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_get_max_threads.restype = C.c_int
        _libs[fast] = L
    return _libs[fast]


class Oracle:
    """One CPU SPH system; phases callable one by one like the reference's z* methods."""

    def __init__(self, pos, vel, box, grid, cell_mode=CELL_MORTON, fast=False):
        self.L = lib(fast)
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        vel = np.ascontiguousarray(vel if vel is not None else np.zeros_like(pos), dtype=np.float32).reshape(-1, 3)
        self.n = pos.shape[0]
        if np.isscalar(box):
            box = (box,) * 3
        if np.isscalar(grid):
            grid = (grid,) * 3
        b = (C.c_float * 3)(*[float(x) for x in box])
        g = (C.c_uint32 * 3)(*[int(x) for x in grid])
        self.s = self.L.orc_create(self.n, b, g, cell_mode)
        if not self.s:
            raise MemoryError("orc_create failed")
        self.L.orc_load(self.s, pos.ctypes.data, vel.ctypes.data)

    def close(self):
        if self.s:
            self.L.orc_destroy(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- raw views (no copies) -------------------------------------------------
    @property
    def particles(self) -> np.ndarray:
        buf = (C.c_char * (self.n * 88)).from_address(self.s.contents.p)
        return np.frombuffer(buf, dtype=PARTICLE_DTYPE)

    @property
    def B(self) -> np.ndarray:
        sz = self.s.contents.b_size
        buf = (C.c_char * (sz * 8)).from_address(self.s.contents.B)
        return np.frombuffer(buf, dtype=GRID_DTYPE)

    @property
    def Bprime(self) -> np.ndarray:
        sz = self.s.contents.bprime_size
        buf = (C.c_char * (sz * 8)).from_address(self.s.contents.Bprime)
        return np.frombuffer(buf, dtype=GRID_DTYPE)

    @property
    def hpos(self) -> np.ndarray:
        buf = (C.c_char * (self.n * 16)).from_address(self.s.contents.hpos)
        return np.frombuffer(buf, dtype=np.float32).reshape(-1, 4)

    # -- phases ----------------------------------------------------------------
    def map_zindex(self): self.L.orc_map_zindex(self.s)
    def sort(self): self.L.orc_sort(self.s)
    def construct_bgrid(self): self.L.orc_construct_bgrid(self.s)
    def construct_grid_array(self): self.L.orc_construct_grid_array(self.s)
    def compute_densities(self): self.L.orc_compute_densities(self.s)
    def compute_densities_n2(self): self.L.orc_compute_densities_n2(self.s)
    def compute_forces(self): self.L.orc_compute_forces(self.s)
    def particle_collisions(self): self.L.orc_particle_collisions(self.s)
    def integrate(self, dt): self.L.orc_integrate(self.s, float(dt))
    def step(self, dt, n=1):
        for _ in range(n):
            self.L.orc_step(self.s, float(dt))

    def apply_order(self, order):
        order = np.ascontiguousarray(order, dtype=np.uint32)
        assert order.shape[0] == self.n
        self.L.orc_apply_order(self.s, order.ctypes.data)

    # -- by original index -------------------------------------------------------
    def by_index(self, field):
        p = self.particles
        out = np.empty_like(p[field])
        out[p["index"]] = p[field]
        return out

    def state(self):
        return dict(pos=self.by_index("position"), vel=self.by_index("velocity"),
                    density=self.by_index("density"), pressure=self.by_index("pressure"))
