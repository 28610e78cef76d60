"""GPU: the z-slab path with the PRODUCT engine (HipEngine -> libsph_hip.so).  Several slabs share the
one GPU of the test box as in-process ranks (LocalComm); the wire protocol, the pack/unpack kernels,
ghost layers, migration and the density halo are the same code the N-GPU run uses over RCCL."""
import threading

import numpy as np
import pytest

from gpufluidsimulator_amd import capi, ic, slab
from slab_oracle_engine import make_case

pytestmark = pytest.mark.gpu
DT = 5e-7


def _run_slabs(world, box, grid, steps, particles=None, lattice=None):
    hub = slab.LocalComm.Hub(world)
    results, errors = [None] * world, []

    def rank_main(r):
        try:
            sim = slab.SlabSimulation(slab.LocalComm(hub, r), lambda cap, gcap, p, z0, z1: slab.HipEngine(cap, gcap, p, z0, z1, 0),
                                      box, grid, particles=particles, lattice=lattice)
            sim.run(DT, steps)
            results[r] = (sim.gather_state(), dict(sim.stats), sim.cuts, sim.engine.n)
            sim.engine.close()
        except BaseException as e:     # noqa: BLE001
            errors.append(e)
            hub.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=900)
    assert not errors, errors
    return results


def _whole_domain(pos, vel, box, grid, steps):
    with capi.Context(pos.shape[0], box=box, grid=grid) as c:
        c.upload(pos, vel)
        c.step(DT, steps)
        return c.download()


@pytest.mark.parametrize("case,world", [("up", 3), ("shear", 4)])
def test_slabs_with_migration_match_whole_domain(case, world):
    pos, vel, box, grid = make_case(case)
    steps = 24
    res = _run_slabs(world, box, grid, steps, particles=(pos, vel))
    st = res[0][0]
    ref = _whole_domain(pos, vel, box, grid, steps)
    assert sum(r[1]["migrants"] for r in res) > 0
    assert sum(r[3] for r in res) == pos.shape[0]
    assert np.abs(st["pos"] - ref["pos"]).max() <= 1e-6 * max(box)
    assert np.abs(st["vel"] - ref["vel"]).max() <= 1e-5 * np.abs(ref["vel"]).max()
    assert np.abs(st["density"] / ref["density"] - 1).max() <= 1e-5


def test_c2_in_four_slabs_matches_whole_domain():
    """BASELINE config 2 (262144 particles) generated slab by slab from the lattice description."""
    cfg = ic.CONFIGS["C2"]
    steps = 3
    res = _run_slabs(4, cfg["box"], cfg["grid"], steps, lattice=cfg["lattice"])
    st, _, cuts, _ = res[0]
    counts = [r[3] for r in res]
    assert sum(counts) == 262144 and max(counts) - min(counts) <= 2 * 64 * 64 * 2, counts
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=True)
    ref = _whole_domain(pos, vel, cfg["box"], cfg["grid"], steps)
    assert np.abs(st["pos"] - ref["pos"]).max() <= 1e-6 * 8.0
    assert np.abs(st["vel"] - ref["vel"]).max() <= 1e-5 * np.abs(ref["vel"]).max()
    assert np.abs(st["density"] / ref["density"] - 1).max() <= 1e-5


def test_weak_scaling_geometry_in_four_slabs():
    """The bench's N-GPU layout in small: a lattice stretched along z in a box stretched along z
    (ic.weak_scaling_config), jitter scaled by the per-GPU box, count-balanced cuts with the last slab
    owning the empty three quarters of the box."""
    cfg = ic.weak_scaling_config(4, per_gpu=(32, 32, 32))
    hub_world, steps = 4, 3
    hub = slab.LocalComm.Hub(hub_world)
    results, errors = [None] * hub_world, []

    def rank_main(r):
        try:
            sim = slab.SlabSimulation(slab.LocalComm(hub, r), lambda cap, gcap, p, z0, z1: slab.HipEngine(cap, gcap, p, z0, z1, 0),
                                      cfg["box"], cfg["grid"], lattice=cfg["lattice"], jitter=True,
                                      jitter_dims=cfg["jitter_dims"])
            sim.run(DT, steps)
            results[r] = (sim.gather_state(), sim.cuts, sim.engine.n)
            sim.engine.close()
        except BaseException as e:     # noqa: BLE001
            errors.append(e)
            hub.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(hub_world)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=900)
    assert not errors, errors
    st, cuts, _ = results[0]
    assert cuts == [0, 16, 32, 48, cfg["grid"][2]]
    assert [r[2] for r in results] == [32 * 32 * 32] * 4
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=True, jitter_dims=cfg["jitter_dims"])
    ref = _whole_domain(pos, vel, cfg["box"], cfg["grid"], steps)
    assert np.abs(st["pos"] - ref["pos"]).max() <= 1e-6 * max(cfg["box"])
    assert np.abs(st["vel"] - ref["vel"]).max() <= 1e-5 * np.abs(ref["vel"]).max()
    assert np.abs(st["density"] / ref["density"] - 1).max() <= 1e-5
