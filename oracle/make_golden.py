#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own CPU code -- TEST INFRASTRUCTURE.

Runs oracle/_ref/sph_ref (the reference's SPH/particleSystem.cpp compiled where
it lies, OMP mode; see oracle/Makefile and oracle/ref_harness.cpp) on seeded
inputs produced by gpufluidsimulator_amd.ic and stores inputs + expected outputs
as small fixtures.  Only runs in the build container (needs /root/reference to
have built oracle/_ref); the fixtures are what travels.

    python oracle/make_golden.py            # all fixtures
    python oracle/make_golden.py c1 random  # a subset
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpufluidsimulator_amd import ic  # noqa: E402
from oracle import refio  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def _phase_arrays(recs, steps):
    out = {}
    for s in steps:
        for k in ("zindex", "order", "sorted_z", "bcells", "bprime", "dens", "force", "coll", "state", "hpos"):
            out[f"s{s}_{k}"] = recs[(k, s)]
    return out


def gen_c1():
    """BASELINE config 1: 16^3 lattice, box 4, grid 64^3, dt 5e-7, 100 steps."""
    cfg = ic.CONFIGS["C1"]
    for tag, jitter in (("c1_lattice", False), ("c1_jitter", True)):
        pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=jitter)
        # phases for the first two steps (exact checks need the reference's tie order)
        recs, _ = refio.run_ref(pos, vel, cfg["box"], cfg["grid"][0], ic.DEFAULT_DT, 2, phases=True)
        arrs = _phase_arrays(recs, (1, 2))
        # the long run: states after 1, 10, 100 steps
        recs, stats = refio.run_ref(pos, vel, cfg["box"], cfg["grid"][0], ic.DEFAULT_DT, 100, dump_steps=(1, 10, 100))
        for s in (1, 10, 100):
            arrs[f"state_{s}"] = recs[("state", s)]
        _save(tag, pos=pos, vel=vel, box=np.float32(cfg["box"]), grid=np.uint32(cfg["grid"]),
              dt=np.float32(ic.DEFAULT_DT), **arrs)


def gen_random():
    """Reference-style CONFIG_RANDOM box (particleSystem.cpp:880-905) made nastier:
    random velocities, a clump of 150 particles inside one cell (more than
    GRID_COMPACT_WIDTH = 32, so B' gets several chunks per cell), particles
    sitting on the walls, and an empty cell 0."""
    box, grid, n = (2.0, 2.0, 2.0), 32, 6000
    pos, vel = ic.random_box(n, box, speed=40.0, fill=0.45)
    pos[:, 0] += np.float32(0.3)            # keep Morton cell 0 empty
    rng = np.random.default_rng(7)
    clump = slice(0, 150)
    pos[clump] = (np.float32([-0.5, -0.5, -0.5]) + rng.uniform(0.002, 0.060, (150, 3))).astype(np.float32)
    wall = slice(150, 250)                     # on / beyond the walls: clamp + damping branch
    pos[wall, 1] = np.float32(-1.0) + rng.uniform(-1e-6, 3e-5, 100).astype(np.float32)
    vel[wall, 1] = np.float32(-30.0)
    pos[250:300, 0] = np.float32(1.0) - np.float32(1e-5)
    vel[250:300, 0] = np.float32(25.0)
    dt = np.float32(2e-6)
    recs, _ = refio.run_ref(pos, vel, box, grid, dt, 4, phases=True)
    arrs = _phase_arrays(recs, (1, 2, 3, 4))
    _save("random_clump", pos=pos, vel=vel, box=np.float32(box), grid=np.uint32((grid,) * 3), dt=dt, **arrs)


def gen_c2():
    """BASELINE config 2: 64^3 lattice, box 8, grid 128^3.  States after 1, 2, 3 and 30 steps;
    stores every 61st particle plus float64 checksums of the full arrays."""
    cfg = ic.CONFIGS["C2"]
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=True)
    recs, stats = refio.run_ref(pos, vel, cfg["box"], cfg["grid"][0], ic.DEFAULT_DT, 30, dump_steps=(1, 2, 3, 30))
    print("c2 reference timing:", stats)
    sample = np.arange(0, pos.shape[0], 61, dtype=np.int64)
    arrs = {}
    for s in (1, 2, 3, 30):
        st = recs[("state", s)]
        arrs[f"state_{s}_sample"] = st[sample]
        arrs[f"state_{s}_sum"] = st.astype(np.float64).sum(axis=0)
        arrs[f"state_{s}_abs_sum"] = np.abs(st.astype(np.float64)).sum(axis=0)
    _save("c2_sample", sample=sample, box=np.float32(cfg["box"]), grid=np.uint32(cfg["grid"]),
          lattice=np.uint32(cfg["lattice"]), dt=np.float32(ic.DEFAULT_DT), **arrs)


def gen_morton():
    """Known answers for coord2zIndex / zIndex2coord from the reference run itself:
    zindex of every C1 particle together with the cell coordinates the hash must
    produce (positions are exact lattice points, so the coordinates are known)."""
    cfg = ic.CONFIGS["C1"]
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=False)
    recs, _ = refio.run_ref(pos, vel, cfg["box"], cfg["grid"][0], ic.DEFAULT_DT, 1, phases=True)
    _save("morton_c1", pos=pos, zindex=recs[("zindex", 1)], box=np.float32(cfg["box"]), grid=np.uint32(cfg["grid"]))


def _gen_flow(tag, lattice, box, grid, runup, lock_steps, phase_steps):
    """Developed flow.  The reference runs the dam break `runup` steps (at dt 5e-7 the first particles change
    cell after ~1000 steps; by 3000-4000 there are wall hits, collisions and a steady stream of cell changes);
    that state -- positions and velocities by creation index -- is stored as the fixture's INPUT.  The reference
    is then restarted from exactly these arrays (so the fixture is self-contained) and stepped `lock_steps` times:
    the state after every step is stored for lockstep tests (upload state k, step once, compare with state k+1:
    chaos has no room to accumulate), per-phase records for the first `phase_steps` steps."""
    pos, vel = ic.dam_break_lattice(lattice, box, jitter=True)
    recs, stats = refio.run_ref(pos, vel, box, grid, ic.DEFAULT_DT, runup, dump_steps=(runup,))
    print(f"{tag}: reference run-up of {runup} steps: {stats['seconds']:.1f} s")
    st = recs[("state", runup)]
    pos0, vel0 = np.ascontiguousarray(st[:, 0:3]), np.ascontiguousarray(st[:, 3:6])
    recs, _ = refio.run_ref(pos0, vel0, box, grid, ic.DEFAULT_DT, phase_steps, phases=True)
    arrs = _phase_arrays(recs, tuple(range(1, phase_steps + 1)))
    recs, _ = refio.run_ref(pos0, vel0, box, grid, ic.DEFAULT_DT, lock_steps, dump_steps=tuple(range(1, lock_steps + 1)))
    for s in range(1, lock_steps + 1):
        arrs[f"state_{s}"] = recs[("state", s)]
    cells = lambda p: np.floor((p + np.float32(box[0] / 2)) / np.float32(box[0]) * np.float32(grid)).astype(np.int64)
    moved = int((cells(recs[("state", lock_steps)][:, 0:3]) != cells(pos0)).any(axis=1).sum())
    print(f"{tag}: {moved} of {pos0.shape[0]} particles change cell within the {lock_steps} stored steps; "
          f"|v|max {np.abs(vel0).max():.1f}, collisions at step 1: {int((recs_coll(arrs) > 0).sum())}")
    _save(tag, pos=pos0, vel=vel0, box=np.float32(box), grid=np.uint32((grid,) * 3), dt=np.float32(ic.DEFAULT_DT),
          runup=np.uint32(runup), lock_steps=np.uint32(lock_steps), phase_steps=np.uint32(phase_steps), **arrs)


def recs_coll(arrs):
    return arrs["s1_coll"][:, 3]


def gen_c1_flow():
    """BASELINE config 1 (16^3 particles, box 4, grid 64^3) after 4000 reference steps."""
    cfg = ic.CONFIGS["C1"]
    _gen_flow("c1_flow", cfg["lattice"], cfg["box"], cfg["grid"][0], 4000, 10, 2)


def gen_d24_flow():
    """A bigger block with an interior (24^3 = 13824 particles, box 4, grid 64^3) after 4000 reference steps."""
    _gen_flow("d24_flow", (24, 24, 24), (4.0, 4.0, 4.0), 64, 4000, 3, 1)


def gen_c2_flow():
    """BASELINE config 2 geometry (64^3 particles, box 8, grid 128^3) in DEVELOPED FLOW: the reference runs the dam
    2600 steps (free fall of 0.09 = 1.5 cell edges; whole lattice layers change cell, the floor layers are compressed
    and collide), the full state is the fixture's input; the reference restarts from it for 2 steps.  Outputs by
    creation index are stored for every 61st particle + float64 checksums of the full arrays (as c2_sample does);
    the collision counts of step 1 -- compared bit for bit -- are stored for EVERY particle (uint8)."""
    cfg = ic.CONFIGS["C2"]
    box, grid, runup, lock = cfg["box"], cfg["grid"][0], 2600, 2
    pos, vel = ic.dam_break_lattice(cfg["lattice"], box, jitter=True)
    recs, stats = refio.run_ref(pos, vel, box, grid, ic.DEFAULT_DT, runup, dump_steps=(runup,), timeout=6 * 3600)
    print(f"c2_flow: reference run-up of {runup} steps: {stats['seconds']:.1f} s")
    st = recs[("state", runup)]
    pos0, vel0 = np.ascontiguousarray(st[:, 0:3]), np.ascontiguousarray(st[:, 3:6])
    sample = np.arange(0, pos0.shape[0], 61, dtype=np.int64)
    arrs = {}

    def put(name, a):
        arrs[name + "_sample"] = a[sample]
        arrs[name + "_sum"] = a.astype(np.float64).sum(axis=0)
        arrs[name + "_abs_sum"] = np.abs(a.astype(np.float64)).sum(axis=0)

    recs, _ = refio.run_ref(pos0, vel0, box, grid, ic.DEFAULT_DT, 1, phases=True)
    for k in ("dens", "force", "coll", "state"):
        put("s1_" + k, recs[(k, 1)])
    arrs["s1_coll_count"] = recs[("coll", 1)][:, 3].astype(np.uint8)
    assert (arrs["s1_coll_count"] == recs[("coll", 1)][:, 3]).all()
    arrs["s1_ncells"] = np.uint32(recs[("bcells", 1)].shape[0])
    arrs["s1_max_cell"] = np.uint32(recs[("bcells", 1)][:, 1].max())
    recs, _ = refio.run_ref(pos0, vel0, box, grid, ic.DEFAULT_DT, lock, dump_steps=tuple(range(1, lock + 1)))
    for s in range(1, lock + 1):
        put(f"state_{s}", recs[("state", s)])
    cells = lambda p: np.floor((p + np.float32(box[0] / 2)) / np.float32(box[0]) * np.float32(grid)).astype(np.int64)
    moved = int((cells(recs[("state", lock)][:, 0:3]) != cells(pos0)).any(axis=1).sum())
    print(f"c2_flow: {moved} of {pos0.shape[0]} particles change cell within the {lock} stored steps; "
          f"|v|max {np.abs(vel0).max():.1f}, particles with collisions at step 1: {int((arrs['s1_coll_count'] > 0).sum())}")
    _save("c2_flow", pos=pos0, vel=vel0, sample=sample, box=np.float32(box), grid=np.uint32(cfg["grid"]),
          dt=np.float32(ic.DEFAULT_DT), runup=np.uint32(runup), lock_steps=np.uint32(lock), moved=np.uint32(moved), **arrs)


GENS = {"c1_flow": gen_c1_flow, "d24_flow": gen_d24_flow, "c1": gen_c1, "random": gen_random, "c2": gen_c2, "morton": gen_morton,
        "c2_flow": gen_c2_flow}   # c2_flow: ~40-60 min of reference CPU time

if __name__ == "__main__":
    if not refio.available():
        sys.exit("oracle/_ref/sph_ref missing: run `make -C oracle ref` in the build container")
    for name in (sys.argv[1:] or list(GENS)):
        GENS[name]()
