"""Initial conditions for the SPH step: dam-break lattices and random boxes.

The reference builds its lattice in ``ParticleSystem::initGrid``
(/root/reference/SPH/particleSystem.cpp:839-874) and calls it from
``reset(CONFIG_GRID)`` (:909-916) with an edge of ``ceil(powf(N, 1/3))`` --
which is n+1 for perfect cubes under glibc -- and glibc ``rand()`` jitter, so
the reference's own ICs are not reproducible across platforms (SURVEY.md
section 7.3-4).  The lattice here keeps the reference's formula
``pos = spacing*i + particleRadius + boxMin + (w*u - w/2)*jitter`` (:855-857),
its index order ``i = x + nx*(y + ny*z)`` (:846) and its jitter amplitude
(``particleRadius*0.01``, :911) but uses an exact (nx, ny, nz) lattice and a
counter-based generator seeded with the reference's seed 1973 (:841), so the
same arrays can be produced bit for bit by numpy here and by the C++ host class.
All arithmetic is single precision, one rounding per operation.
"""
from __future__ import annotations

import numpy as np

# physics constants of the reference (SPH/particles_kernel.cuh:20-33,
# SPH/particleSystem.cpp:51, SPH/particles.cpp:88)
PARTICLE_RADIUS = np.float32(1.0 / 64.0)
SPACING = np.float32(2.0) * PARTICLE_RADIUS
SMOOTHING_H = np.float32(0.1)
DEFAULT_DT = np.float32(0.0000005)
SEED = 1973


def next_pow2(x: int) -> int:
    """``nextPow2`` of SPH/particleSystem.h:34-43."""
    x = int(x) - 1
    for s in (1, 2, 4, 8, 16):
        x |= x >> s
    return (x + 1) & 0xFFFFFFFF


def grid_dim_for_box(edge: float) -> int:
    """The reference's grid formula (particleSystem.cpp:46) applied to the box
    edge instead of the BOX_SIZE macro: ``nextPow2((uint)(edge/(0.66666f*h)))``.
    edge 2 -> 32, 4 -> 64, 8 -> 128, 32 -> 512, 64 -> 1024."""
    q = np.float32(edge) / (np.float32(0.66666) * SMOOTHING_H)
    return next_pow2(int(q))


def _hash_u32(x: np.ndarray) -> np.ndarray:
    """32-bit integer finaliser (lowbias32); the C++ twin is ``sph_ic_hash``."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def uniform01(counter: np.ndarray, stream: int, seed: int = SEED) -> np.ndarray:
    """Counter-based uniform in [0,1) with 24 random bits (exact in fp32)."""
    c = counter.astype(np.uint32)
    with np.errstate(over="ignore"):
        k = _hash_u32(c * np.uint32(3) + np.uint32(stream) + np.uint32(seed) * np.uint32(0x9E3779B9))
        k = _hash_u32(k ^ np.uint32(0x85EBCA6B))
    return (k >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def dam_break_lattice(lattice, box, jitter: bool = True, start: int = 0, count: int | None = None,
                      jitter_dims=None):
    """Positions/velocities of an (nx, ny, nz) lattice in the min corner of a box
    of dimensions ``box`` centred on the origin.

    Returns (pos[n,3] float32, vel[n,3] float32) for particle indices
    ``start .. start+count`` (default: all), index = x + nx*(y + ny*z).
    ``jitter_dims`` replaces the box dimensions in the jitter term only (the reference scales the
    jitter with the box, which on a box stretched along z for weak scaling would exceed the spacing).
    """
    nx, ny, nz = (int(v) for v in lattice)
    w = np.asarray(box, dtype=np.float32).reshape(3)
    wj = w if jitter_dims is None else np.asarray(jitter_dims, dtype=np.float32).reshape(3)
    n_total = nx * ny * nz
    if count is None:
        count = n_total - start
    idx = np.arange(start, start + count, dtype=np.int64)
    ix = (idx % nx).astype(np.float32)
    iy = ((idx // nx) % ny).astype(np.float32)
    iz = (idx // (nx * ny)).astype(np.float32)
    box_min = -w / np.float32(2.0)
    pos = np.empty((count, 3), dtype=np.float32)
    jit = PARTICLE_RADIUS * np.float32(0.01)
    cnt = idx.astype(np.uint32)
    for a, ia in enumerate((ix, iy, iz)):
        base = (SPACING * ia + PARTICLE_RADIUS) + box_min[a]
        if jitter:
            u = uniform01(cnt, a)
            base = base + (wj[a] * u - wj[a] / np.float32(2.0)) * jit
        pos[:, a] = base
    vel = np.zeros((count, 3), dtype=np.float32)
    return pos, vel


def random_box(n: int, box, speed: float = 0.0, seed: int = SEED, fill: float = 1.0):
    """``reset(CONFIG_RANDOM)`` analogue (particleSystem.cpp:880-905): uniform
    positions ``w*u - w/2`` (optionally confined to the lowest ``fill`` fraction
    of each axis) and, unlike the reference (zero), optional random velocities
    so that the collision and wall branches are exercised by small tests."""
    w = np.asarray(box, dtype=np.float32).reshape(3)
    cnt = np.arange(n, dtype=np.uint32)
    pos = np.empty((n, 3), dtype=np.float32)
    vel = np.zeros((n, 3), dtype=np.float32)
    for a in range(3):
        u = uniform01(cnt, a, seed)
        pos[:, a] = w[a] * (u * np.float32(fill)) - w[a] / np.float32(2.0)
        if speed:
            v = uniform01(cnt, 3 + a, seed)
            vel[:, a] = (v - np.float32(0.5)) * np.float32(2.0 * speed)
    return pos, vel


# BASELINE.json configs: name -> (lattice, box, grid, steps)
CONFIGS = {
    "C1": dict(lattice=(16, 16, 16), box=(4.0, 4.0, 4.0), grid=(64, 64, 64), steps=100),
    "C2": dict(lattice=(64, 64, 64), box=(8.0, 8.0, 8.0), grid=(128, 128, 128), steps=20),
    "C3": dict(lattice=(256, 256, 256), box=(32.0, 32.0, 32.0), grid=(512, 512, 512), steps=20),
    # BASELINE config 4: 67,108,864 particles in total, cut into z-slabs over 2 -> 4 -> 8 GPUs (strong scaling);
    # long axis = z, the slab axis
    "C4": dict(lattice=(256, 512, 512), box=(64.0, 64.0, 64.0), grid=(1024, 1024, 1024), steps=20),
    # BASELINE config 5: 2^27 = 134,217,728 particles ("C5 n = 512, L = 64"); run with
    # sph_set_precision(MIXED_F16) -- fp32 positions, fp16 neighbour accumulators.  ~32 GB of HBM: fits one MI355X.
    "C5": dict(lattice=(512, 512, 512), box=(64.0, 64.0, 64.0), grid=(1024, 1024, 1024), steps=20),
}


def weak_scaling_config(n_gpus: int, per_gpu=(256, 256, 256)):
    """C3 replicated along z: every rank owns ``per_gpu`` lattice layers.  The box
    and grid grow along z only (cell edge stays 0.0625), so a rank's slab looks
    like C3 with open z faces."""
    nx, ny, nz = per_gpu
    gx = grid_dim_for_box(nx / 8.0)
    gy = grid_dim_for_box(ny / 8.0)
    gz1 = grid_dim_for_box(nz / 8.0)
    return dict(lattice=(nx, ny, nz * n_gpus), box=(gx / 16.0, gy / 16.0, gz1 * n_gpus / 16.0),
                grid=(gx, gy, gz1 * n_gpus), steps=20, jitter_dims=(gx / 16.0, gy / 16.0, gz1 / 16.0))
