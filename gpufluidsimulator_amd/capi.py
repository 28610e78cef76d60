"""ctypes binding of the C ABI in include/sph_hip.h (libsph_hip.so).

This is plumbing only: every compute call lands in the hand-written HIP library.  There is
no CPU fallback -- a missing library or a missing gfx950 device raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPH_HIP_LIB", os.path.join(HERE, "libsph_hip.so"))

PHASES = ("z-index", "sort", "b-grid", "dens", "force", "collision", "integrate")
HALO_RECORD_FLOATS = 8


class SphError(RuntimeError):
    pass


# int exchange(void* self, int tag, const void* send_lo, size_t, void* recv_lo, size_t, const void* send_hi, size_t,
#              void* recv_hi, size_t, void* hip_stream)   -- `sph_transport.exchange` of include/sph_hip.h
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                          C.c_void_p, C.c_size_t, C.c_void_p)


class Transport(C.Structure):
    """`sph_transport` of include/sph_hip.h."""
    _fields_ = [("self", C.c_void_p), ("exchange", EXCHANGE_FN), ("host_buffers", C.c_int),
                ("abort", C.CFUNCTYPE(None, C.c_void_p))]


class Params(C.Structure):
    """`sph_params` of include/sph_hip.h."""
    _fields_ = [("box_min", C.c_float * 3), ("box_max", C.c_float * 3), ("grid", C.c_uint32 * 3),
                ("h", C.c_float), ("mass", C.c_float), ("rest_density", C.c_float), ("gas_constant", C.c_float),
                ("viscosity", C.c_float), ("gravity_y", C.c_float), ("wall_eps", C.c_float),
                ("wall_damping", C.c_float), ("restitution", C.c_float), ("collision_param", C.c_float),
                ("particle_radius", C.c_float)]


# name -> (restype, argtypes); also the list the symbol-export test walks
_P = C.c_void_p
_U32 = C.c_uint32
SIGNATURES = {
    "sph_abi_version": (C.c_int, []),
    "sph_last_error": (C.c_char_p, []),
    "sph_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sph_select_device": (C.c_int, [C.c_int]),
    "sph_selected_device": (C.c_int, []),
    "sph_default_params": (None, [C.POINTER(Params), C.POINTER(C.c_float), C.POINTER(_U32)]),
    "sph_grid_dim_for_edge": (_U32, [C.c_float, C.c_float]),
    "sph_create": (C.c_int, [C.POINTER(_P), C.c_int, _U32, C.POINTER(Params)]),
    "sph_create_slab": (C.c_int, [C.POINTER(_P), C.c_int, _U32, C.POINTER(Params), _U32, _U32, _U32]),
    "sph_create_slab_layers": (C.c_int, [C.POINTER(_P), C.c_int, _U32, C.POINTER(Params), _U32, _U32, _U32, _U32]),
    "sph_ghost_layers": (_U32, [_P]),
    "sph_slab_set_protocol": (C.c_int, [_P, C.c_int]),
    "sph_slab_protocol": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "sph_destroy": (None, [_P]),
    "sph_set_stream": (C.c_int, [_P, _P]),
    "sph_set_params": (C.c_int, [_P, C.POINTER(Params)]),
    "sph_get_params": (C.c_int, [_P, C.POINTER(Params)]),
    "sph_sync": (C.c_int, [_P]),
    "sph_num_particles": (_U32, [_P]),
    "sph_capacity": (_U32, [_P]),
    "sph_upload": (C.c_int, [_P, _U32, _P, _P, _P]),
    "sph_set_by_index": (C.c_int, [_P, _U32, _U32, C.c_void_p, C.c_void_p]),
    "sph_reset_lattice": (C.c_int, [_P, C.POINTER(_U32), C.c_int, C.POINTER(C.c_float), C.c_uint64, _U32]),
    "sph_download": (C.c_int, [_P, _U32, _U32, _P, _P, _P, _P]),
    "sph_download_owned": (C.c_int, [_P, _P, _P, _P]),
    "sph_download_forces": (C.c_int, [_P, _U32, _U32, _P, _P, _P, _P]),
    "sph_snapshot_save": (C.c_int, [_P, C.c_char_p]),
    "sph_snapshot_load": (C.c_int, [_P, C.c_char_p]),
    "sph_snapshot_info": (C.c_int, [C.c_char_p, C.POINTER(_U32), C.POINTER(Params)]),
    "sph_positions_dev": (C.c_int, [_P, C.POINTER(_P)]),
    "sph_download_positions4": (C.c_int, [_P, _P]),
    "sph_get_keys": (C.c_int, [_P, _P]),
    "sph_get_order": (C.c_int, [_P, _P]),
    "sph_get_cell_range": (C.c_int, [_P, _U32, C.POINTER(_U32), C.POINTER(_U32)]),
    "sph_get_cells": (C.c_int, [_P, _U32, _P, _P, _P]),
    "sph_cell_key": (_U32, [_P, _U32, _U32, _U32]),
    "sph_hash": (C.c_int, [_P]),
    "sph_sort": (C.c_int, [_P]),
    "sph_build_cells": (C.c_int, [_P]),
    "sph_density": (C.c_int, [_P]),
    "sph_force": (C.c_int, [_P]),
    "sph_collide": (C.c_int, [_P]),
    "sph_integrate": (C.c_int, [_P, C.c_float]),
    "sph_step": (C.c_int, [_P, C.c_float, _U32]),
    "sph_step_phased": (C.c_int, [_P, C.c_float, _U32]),
    "sph_force_collide_integrate": (C.c_int, [_P, C.c_float]),
    "sph_timing_enable": (C.c_int, [_P, C.c_int]),
    "sph_timing_get": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(_U32)]),
    "sph_timing_reset": (C.c_int, [_P]),
    "sph_last_sort_skipped": (C.c_int, [_P]),
    "sph_sort_stats": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint64)]),
    "sph_set_sort_mode": (C.c_int, [_P, C.c_int]),
    "sph_set_direct_hull": (C.c_int, [_P, C.c_uint32]),
    "sph_set_pair_small_launch": (C.c_int, [_P, C.c_uint32]),
    "sph_set_block_order": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint32]),
    "sph_sort_forms": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "sph_test_trust_mover_hint": (C.c_int, [_P]),
    "sph_set_precision": (C.c_int, [_P, C.c_int]),
    "sph_get_precision": (C.c_int, [_P]),
    "sph_migrants_count": (C.c_int, [_P, C.POINTER(_U32)]),
    "sph_slab_counts": (C.c_int, [_P, C.POINTER(_U32)]),
    "sph_migrants_pack": (C.c_int, [_P, C.POINTER(_P), _U32]),
    "sph_migrants_append": (C.c_int, [_P, _P, _U32]),
    "sph_halo_count": (C.c_int, [_P, C.POINTER(_U32)]),
    "sph_halo_pack": (C.c_int, [_P, C.POINTER(_P), _U32]),
    "sph_halo_pack_counts": (C.c_int, [_P, C.POINTER(_P), _U32, C.POINTER(_U32)]),
    "sph_halo_unpack": (C.c_int, [_P, _P, _U32, _P, _U32]),
    "sph_halo_pack_density": (C.c_int, [_P, C.POINTER(_P), _U32]),
    "sph_halo_unpack_density": (C.c_int, [_P, _P, _P]),
    "sph_layer_histogram": (C.c_int, [_P, _P, _U32]),
    "sph_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "sph_rccl_transport_create": (C.c_int, [C.POINTER(C.POINTER(Transport)), C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "sph_rccl_transport_destroy": (None, [C.POINTER(Transport)]),
    "sph_rccl_transport_selftest": (C.c_int, [C.POINTER(Transport), C.c_size_t]),
    "sph_rccl_transport_info": (C.c_int, [C.POINTER(Transport), C.POINTER(C.c_int)]),
    "sph_slab_ping": (C.c_int, [_P, C.c_size_t, _U32, C.POINTER(C.c_double)]),
    "sph_slab_recut": (C.c_int, [_P, _U32, _U32]),
    "sph_slab_recut_stats": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "sph_slab_set_early_force": (C.c_int, [_P, C.c_int]),
    "sph_slab_early_force_stats": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "sph_slab_timing_enable": (C.c_int, [_P, C.c_int]),
    "sph_slab_timing_reset": (C.c_int, [_P]),
    "sph_slab_timing_get": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "sph_slab_test_raise_flag": (C.c_int, [_P, C.c_int]),
    "sph_slab_in_place_merges": (C.c_uint64, [C.c_void_p]),
    "sph_slab_exchanges": (C.c_uint64, [C.c_void_p]),
    "sph_slab_failed": (C.c_int, [C.c_void_p]),
    "sph_slab_create": (C.c_int, [C.POINTER(_P), _P, C.c_int, C.c_int, C.POINTER(Transport), _U32]),
    "sph_slab_destroy": (None, [_P]),
    "sph_slab_step": (C.c_int, [_P, C.c_float, _U32]),
    "sph_slab_sync": (C.c_int, [_P]),
    "sph_slab_stats": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "sph_slab_counters": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "sph_slab_set_wait_timeout": (C.c_int, [_P, C.c_double]),
    "sph_local_hub_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int]),
    "sph_local_hub_destroy": (None, [_P]),
    "sph_local_hub_set_timeout": (C.c_int, [_P, C.c_double]),
    "sph_local_transport_create": (C.c_int, [C.POINTER(C.POINTER(Transport)), _P, C.c_int]),
    "sph_local_transport_destroy": (None, [C.POINTER(Transport)]),
    "sph_loop_transport_create": (C.c_int, [C.POINTER(C.POINTER(Transport)), C.c_float, C.c_double, C.c_double]),
    "sph_loop_transport_destroy": (None, [C.POINTER(Transport)]),
    # include/particleSystem.h: host-only twins of ic.py (used by the C++ class's reset())
    "sph_ic_dam_break": (None, [C.POINTER(_U32), C.POINTER(C.c_float), C.c_int, C.c_uint64, C.c_uint64, _P, _P]),
    "sph_ic_random_box": (None, [C.c_uint64, C.POINTER(C.c_float), C.c_float, _U32, C.c_float, _P, _P]),
}

_lib = None
_torch_first = None     # was torch already imported when libsph_hip.so (and with it /opt/rocm's HIP runtime) was loaded?


def load():
    """dlopen libsph_hip.so (building it first if it is missing) and type its entry points.

    Library order matters in a process that also uses PyTorch: the torch wheel bundles its own HIP/HSA
    runtime and loads it by path.  If libsph_hip.so (linked against /opt/rocm) comes first, torch later
    brings up a second runtime, which finds no device.  Import torch BEFORE the first call of this
    function; then libsph_hip.so binds to the runtime torch has already loaded (same SONAMEs)."""
    global _lib, _torch_first
    if _lib is None:
        import sys
        _torch_first = "torch" in sys.modules
        if not os.path.exists(LIB_PATH):
            from . import build as _b
            _b.build()
        lib = C.CDLL(LIB_PATH)
        lenient = bool(os.environ.get("SPH_HIP_LIB_ALLOW_MISSING"))     # A/B runs against a library of an EARLIER round (profiles/scripts)
        for name, (res, args) in SIGNATURES.items():
            if lenient and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)          # AttributeError = missing export: fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.sph_abi_version() != 2:
            raise SphError("libsph_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def _check(rc):
    if rc < 0:
        raise SphError(f"libsph_hip error {rc}: {load().sph_last_error().decode()}")
    return rc


def default_params(box, grid) -> Params:
    p = Params()
    load().sph_default_params(C.byref(p), (C.c_float * 3)(*[float(b) for b in box]),
                              (C.c_uint32 * 3)(*[int(g) for g in grid]))
    return p


def device_count():
    flag = C.c_int(0)
    n = load().sph_device_count(C.byref(flag))
    return n, bool(flag.value)


class LocalHub:
    """`sph_local_hub`: the rendezvous of the device-to-device transport between slabs of one process (one GPU)."""

    def __init__(self, world, device=0, timeout_s=None):
        h = _P()
        _check(load().sph_local_hub_create(C.byref(h), int(world), int(device)))
        self.h = h
        if timeout_s:
            _check(load().sph_local_hub_set_timeout(self.h, float(timeout_s)))

    def transport(self, rank):
        t = C.POINTER(Transport)()
        _check(load().sph_local_transport_create(C.byref(t), self.h, int(rank)))
        return t

    def close(self):
        if getattr(self, "h", None):
            load().sph_local_hub_destroy(self.h)
            self.h = None


def _f32(a, cols):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(-1, cols)


class Context:
    """One `sph_ctx`: the device state of a particle system (or of one z-slab of it)."""

    def __init__(self, capacity, box=None, grid=None, params: Params | None = None, device=0,
                 slab=None, ghost_capacity=0, ghost_layers=1):
        self.L = load()
        self.params = params if params is not None else default_params(box, grid)
        h = _P()
        if slab is None:
            _check(self.L.sph_create(C.byref(h), device, int(capacity), C.byref(self.params)))
        else:       # ghost_layers = 2: what the one-message slab step needs (sph_slab_set_protocol)
            _check(self.L.sph_create_slab_layers(C.byref(h), device, int(capacity), C.byref(self.params),
                                                 int(slab[0]), int(slab[1]), int(ghost_capacity), int(ghost_layers)))
        self.h = h
        self.capacity = int(capacity)
        self.index_base = 0
        self.index_count = int(capacity)

    def close(self):
        if getattr(self, "h", None):
            self.L.sph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- state ---------------------------------------------------------------------------------
    @property
    def n(self):
        return int(self.L.sph_num_particles(self.h))

    def set_stream(self, stream_handle):
        _check(self.L.sph_set_stream(self.h, _P(stream_handle)))

    def set_params(self, params: Params):
        _check(self.L.sph_set_params(self.h, C.byref(params)))
        self.params = params

    def upload(self, pos, vel=None, index=None):
        pos = _f32(pos, 3)
        n = pos.shape[0]
        vel = _f32(vel, 3) if vel is not None else None
        idx = np.ascontiguousarray(index, dtype=np.uint32) if index is not None else None
        _check(self.L.sph_upload(self.h, n, pos.ctypes.data, vel.ctypes.data if vel is not None else None,
                                 idx.ctypes.data if idx is not None else None))

    def set_by_index(self, first_index, pos=None, vel=None):
        """Overwrite position and/or velocity of the particles with creation indices first_index .. (device side)."""
        pos = None if pos is None else np.ascontiguousarray(pos, dtype=np.float32)
        vel = None if vel is None else np.ascontiguousarray(vel, dtype=np.float32)
        count = (pos if pos is not None else vel).shape[0]
        assert vel is None or pos is None or vel.shape == pos.shape
        _check(self.L.sph_set_by_index(self.h, int(first_index), int(count),
                                       None if pos is None else pos.ctypes.data, None if vel is None else vel.ctypes.data))

    def reset_lattice(self, lattice, jitter=True, jitter_dims=None, start=0, count=None):
        """Dam-break lattice generated on the device (bit-identical to ic.dam_break_lattice)."""
        lat = (_U32 * 3)(*[int(v) for v in lattice])
        total = int(lattice[0]) * int(lattice[1]) * int(lattice[2])
        count = total - start if count is None else count
        jd = (C.c_float * 3)(*[float(v) for v in jitter_dims]) if jitter_dims is not None else None
        _check(self.L.sph_reset_lattice(self.h, lat, 1 if jitter else 0, jd, int(start), int(count)))

    def download(self, index_base=0, count=None, want=("pos", "vel", "density", "pressure")):
        """State by creation index; rows of particles this context does not own stay NaN."""
        count = self.index_count if count is None else count
        out = {}
        for k in want:
            out[k] = np.full((count, 3) if k in ("pos", "vel") else (count,), np.nan, dtype=np.float32)
        ptr = lambda k: out[k].ctypes.data if k in out else None
        _check(self.L.sph_download(self.h, int(index_base), int(count), ptr("pos"), ptr("vel"), ptr("density"),
                                   ptr("pressure")))
        return out

    def download_owned(self):
        """(pos[n,3], vel[n,3], index[n]) of the owned particles in slot order."""
        n = self.n
        pos, vel, idx = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty(n, np.uint32)
        _check(self.L.sph_download_owned(self.h, pos.ctypes.data, vel.ctypes.data, idx.ctypes.data))
        return pos, vel, idx

    def download_forces(self, index_base=0, count=None, force=True, collision=True):
        count = self.index_count if count is None else count
        out = {}
        if force:
            out["fpress"] = np.full((count, 3), np.nan, dtype=np.float32)
            out["fvisc"] = np.full((count, 3), np.nan, dtype=np.float32)
        if collision:
            out["dv"] = np.full((count, 3), np.nan, dtype=np.float32)
            out["count"] = np.full((count,), -1, dtype=np.int32)
        ptr = lambda k: out[k].ctypes.data if k in out else None
        _check(self.L.sph_download_forces(self.h, int(index_base), int(count), ptr("fpress"), ptr("fvisc"), ptr("dv"),
                                          ptr("count")))
        return out

    def save(self, path):
        _check(self.L.sph_snapshot_save(self.h, os.fsencode(path)))

    def load_snapshot(self, path):
        _check(self.L.sph_snapshot_load(self.h, os.fsencode(path)))

    @staticmethod
    def snapshot_info(path):
        n, p = _U32(0), Params()
        _check(load().sph_snapshot_info(os.fsencode(path), C.byref(n), C.byref(p)))
        return int(n.value), p

    def positions4(self):
        out = np.empty((self.capacity, 4), dtype=np.float32)
        _check(self.L.sph_download_positions4(self.h, out.ctypes.data))
        return out

    def positions_dev(self):
        p = _P()
        _check(self.L.sph_positions_dev(self.h, C.byref(p)))
        return p.value

    def keys(self):
        out = np.empty(self.n, dtype=np.uint32)
        _check(self.L.sph_get_keys(self.h, out.ctypes.data))
        return out

    def order(self):
        out = np.empty(self.n, dtype=np.uint32)
        _check(self.L.sph_get_order(self.h, out.ctypes.data))
        return out

    def cells(self, max_cells=None):
        max_cells = self.n + 16 if max_cells is None else max_cells
        k = np.empty(max_cells, dtype=np.uint32)
        s = np.empty(max_cells, dtype=np.uint32)
        c = np.empty(max_cells, dtype=np.uint32)
        m = _check(self.L.sph_get_cells(self.h, max_cells, k.ctypes.data, s.ctypes.data, c.ctypes.data))
        m = min(m, max_cells)
        return k[:m], s[:m], c[:m]

    def cell_range(self, cell):
        """(start, end) of one cell of the table, relative to the first installed slot; (0, 0) = empty."""
        a, b = _U32(), _U32()
        _check(self.L.sph_get_cell_range(self.h, int(cell), C.byref(a), C.byref(b)))
        return a.value, b.value

    def cell_key(self, x, y, z):
        return int(self.L.sph_cell_key(self.h, int(x), int(y), int(z)))

    # -- phases ---------------------------------------------------------------------------------
    def hash(self): _check(self.L.sph_hash(self.h))
    def sort(self): _check(self.L.sph_sort(self.h))
    def build_cells(self): _check(self.L.sph_build_cells(self.h))
    def density(self): _check(self.L.sph_density(self.h))
    def force(self): _check(self.L.sph_force(self.h))
    def collide(self): _check(self.L.sph_collide(self.h))
    def integrate(self, dt): _check(self.L.sph_integrate(self.h, float(dt)))
    def step(self, dt, n=1): _check(self.L.sph_step(self.h, float(dt), int(n)))
    def step_phased(self, dt, n=1): _check(self.L.sph_step_phased(self.h, float(dt), int(n)))
    def sync(self): _check(self.L.sph_sync(self.h))

    # -- timing ---------------------------------------------------------------------------------
    def timing(self, on=True): _check(self.L.sph_timing_enable(self.h, 1 if on else 0))
    def timing_reset(self): _check(self.L.sph_timing_reset(self.h))

    def sort_skipped(self):
        """True if the last sort found no particle in a new cell and did nothing (no synchronisation)."""
        return bool(self.L.sph_last_sort_skipped(self.h))

    def sort_stats(self):
        """{'sorts', 'merges', 'skips', 'last_movers', 'movers_total'}: how often the sort took the merge path and
        how many particles changed cell (see sph_hip.h)."""
        a, b, k, m, t = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint64()
        _check(self.L.sph_sort_stats(self.h, C.byref(a), C.byref(b), C.byref(k), C.byref(m), C.byref(t)))
        return {"sorts": a.value, "merges": b.value, "skips": k.value, "last_movers": m.value, "movers_total": t.value}

    def set_precision(self, mixed_f16=False):
        """False: fp32 (the reference's precision); True: BASELINE config 5 -- fp32 state, packed-fp16 pair arithmetic
        and per-row accumulators in the density / force traversals."""
        _check(self.L.sph_set_precision(self.h, 1 if mixed_f16 else 0))

    def set_sort_mode(self, merge=True):
        """merge=False/0: full radix sort every step (the SPH_SORT_MERGE=0 behaviour); True/1: merge while few
        particles change cell; 2: merge whatever the count (tests)."""
        _check(self.L.sph_set_sort_mode(self.h, int(merge)))

    def trust_mover_hint(self):
        """Test hook: after set_by_index / upload the next movers' sorts launch both forms; this takes that back."""
        _check(self.L.sph_test_trust_mover_hint(self.h))

    def sort_forms(self):
        """Movers' sorts launched as (both forms, one-block sort alone, multi-block passes alone)."""
        out = (C.c_uint64 * 3)()
        _check(self.L.sph_sort_forms(self.h, out))
        return tuple(int(v) for v in out)

    def set_block_order(self, xcd=True, ztile=True, strip_blocks_log2=4):
        """Order in which the pair kernels' workgroups take the slots (performance only; see sph_set_block_order)."""
        _check(self.L.sph_set_block_order(self.h, int(bool(xcd)), int(bool(ztile)), int(strip_blocks_log2)))

    def set_direct_hull(self, slots=512):
        """Rows of the neighbour passes whose staged hull would exceed `slots` are read straight from global memory
        (0: all of them, 0xFFFFFFFF: none); same bits either way."""
        _check(self.L.sph_set_direct_hull(self.h, int(slots) & 0xFFFFFFFF))

    def set_pair_small_launch(self, slots):
        """A context with fewer than `slots` owned particles launches the neighbour passes in 128-thread workgroups (0: never,
        0xFFFFFFFF: always; default 524288); same bits either way."""
        _check(self.L.sph_set_pair_small_launch(self.h, int(slots) & 0xFFFFFFFF))

    def timing_get(self):
        ms = (C.c_float * len(PHASES))()
        steps = _U32(0)
        _check(self.L.sph_timing_get(self.h, ms, C.byref(steps)))
        return {PHASES[k]: float(ms[k]) for k in range(len(PHASES))}, int(steps.value)

    # -- slab halo (device pointers: ints, e.g. torch.Tensor.data_ptr()) ----------------------------
    def _pair(self, fn, *a):
        cnt = (_U32 * 2)()
        _check(fn(self.h, cnt, *a))
        return int(cnt[0]), int(cnt[1])

    def slab_counts(self):
        cnt = (_U32 * 4)()
        _check(self.L.sph_slab_counts(self.h, cnt))
        return tuple(int(v) for v in cnt)

    def force_collide_integrate(self, dt): _check(self.L.sph_force_collide_integrate(self.h, float(dt)))

    def migrants_count(self): return self._pair(self.L.sph_migrants_count)
    def halo_count(self): return self._pair(self.L.sph_halo_count)

    def migrants_pack(self, buf_lo, buf_hi, capacity):
        _check(self.L.sph_migrants_pack(self.h, (_P * 2)(buf_lo, buf_hi), int(capacity)))

    def migrants_append(self, buf, n):
        _check(self.L.sph_migrants_append(self.h, _P(buf), int(n)))

    def halo_pack(self, buf_lo, buf_hi, capacity, counts=None):
        if counts is None:
            _check(self.L.sph_halo_pack(self.h, (_P * 2)(buf_lo, buf_hi), int(capacity)))
        else:
            _check(self.L.sph_halo_pack_counts(self.h, (_P * 2)(buf_lo, buf_hi), int(capacity),
                                               (_U32 * 2)(int(counts[0]), int(counts[1]))))

    def halo_unpack(self, lo, n_lo, hi, n_hi):
        _check(self.L.sph_halo_unpack(self.h, _P(lo), int(n_lo), _P(hi), int(n_hi)))

    def halo_pack_density(self, buf_lo, buf_hi, capacity):
        _check(self.L.sph_halo_pack_density(self.h, (_P * 2)(buf_lo, buf_hi), int(capacity)))

    def halo_unpack_density(self, lo, hi):
        _check(self.L.sph_halo_unpack_density(self.h, _P(lo), _P(hi)))

    def layer_histogram(self):
        n = int(self.params.grid[2])
        out = np.zeros(n, dtype=np.uint32)
        _check(self.L.sph_layer_histogram(self.h, out.ctypes.data, n))
        return out
