"""GPU: BASELINE config 5 -- fp32 positions with fp16 neighbour accumulators (sph_set_precision MIXED_F16).

The density pass does its pair arithmetic in packed fp16 (two candidates per lane-instruction) on coordinates
relative to a wave-local reference in units of h, sums the NORMALISED kernel per row in fp16 and the rows in fp32.
fp16 carries 11 significant bits: a coordinate up to 4 h away from the reference is rounded to 2^-9 h, r^2/h^2 to a
few 1e-3.  Stated tolerance ("mixed"): density and pressure within 2e-2 of the fp32 value for
every particle and within 4e-3 rms; pressure force within 1e-2 of the largest force; everything that does not depend
on the density (cell keys, order, collision counts, delta-v) bit-identical to the fp32 path; after 20 free steps
velocities within 5e-3 of |v|max and positions within 2e-6 of the box edge of the fp32 run."""
import numpy as np
import pytest

from conftest import load_golden
from gpufluidsimulator_amd import capi
from oracle import oracle

pytestmark = pytest.mark.gpu

RHO_MAX, RHO_RMS, FORCE_TOL, VEL_TOL_20, POS_TOL_20 = 2e-2, 4e-3, 1e-2, 5e-3, 2e-6


@pytest.mark.parametrize("name", ["c1_jitter", "c1_flow", "d24_flow", "random_clump"])
def test_mixed_density_against_the_oracle_and_fp32(name):
    g = load_golden(name)
    o = oracle.Oracle(g["pos"], g["vel"], g["box"], g["grid"], oracle.CELL_LINEAR)
    o.map_zindex(); o.sort(); o.construct_bgrid(); o.compute_densities()
    want = o.by_index("density")
    o.close()
    res = {}
    with capi.Context(g["pos"].shape[0], box=g["box"], grid=g["grid"]) as c:
        for mixed in (False, True):
            c.set_precision(mixed)
            c.upload(g["pos"], g["vel"])
            c.hash(); c.sort(); c.build_cells(); c.density(); c.force(); c.collide()
            res[mixed] = (c.download(want=("density", "pressure")), c.download_forces(), c.keys(), c.order())
    (s32, f32, k32, o32), (s16, f16, k16, o16) = res[False], res[True]
    rel = s16["density"] / want - 1
    assert np.abs(rel).max() <= RHO_MAX and np.sqrt(np.mean(rel ** 2)) <= RHO_RMS
    assert np.abs(s16["pressure"] - s32["pressure"]).max() <= RHO_MAX * np.abs(s32["pressure"]).max()
    fscale = max(np.abs(f32["fpress"]).max(), np.abs(f32["fvisc"]).max())
    assert np.abs(f16["fpress"] - f32["fpress"]).max() <= FORCE_TOL * fscale
    assert np.abs(f16["fvisc"] - f32["fvisc"]).max() <= FORCE_TOL * fscale
    # nothing that does not depend on the density may change
    assert np.array_equal(k16, k32) and np.array_equal(o16, o32)
    assert np.array_equal(f16["count"], f32["count"]) and np.array_equal(f16["dv"], f32["dv"])


@pytest.mark.parametrize("name", ["c1_jitter", "c1_flow"])
def test_mixed_free_run_stays_close_to_fp32(name):
    g = load_golden(name)
    dt, box = float(g["dt"]), float(g["box"].max())
    out = {}
    with capi.Context(g["pos"].shape[0], box=g["box"], grid=g["grid"]) as c:
        for mixed in (False, True):
            c.set_precision(mixed)
            c.upload(g["pos"], g["vel"])
            c.step(dt, 20)
            out[mixed] = c.download()
        # fused and phase-by-phase steps agree in mixed mode too
        c.upload(g["pos"], g["vel"])
        c.step_phased(dt, 20)
        ph = c.download()
    a, b = out[False], out[True]
    assert np.abs(b["vel"] - a["vel"]).max() <= VEL_TOL_20 * np.abs(a["vel"]).max()
    assert np.abs(b["pos"] - a["pos"]).max() <= POS_TOL_20 * box
    assert np.isfinite(b["density"]).all()
    assert np.abs(ph["vel"] - b["vel"]).max() <= 1e-5 * np.abs(b["vel"]).max()
    assert np.array_equal(ph["density"], b["density"])


def test_precision_switch_is_validated():
    with capi.Context(64, box=(2, 2, 2), grid=(32, 32, 32)) as c:
        assert c.L.sph_get_precision(c.h) == 0
        c.set_precision(True)
        assert c.L.sph_get_precision(c.h) == 1
        assert c.L.sph_set_precision(c.h, 7) < 0


def test_mixed_density_on_waves_whose_particles_are_far_apart():
    """Round 5: the packed-fp16 arithmetic works on coordinates relative to the wave's first particle, which is 2^-9 h only
    while the wave's 64 sorted particles lie within a few h of each other.  A wave that straddles the end of an x-row (its
    second half starts at the other side of the fluid) or holds the scattered particles of a nearly empty layer had
    densities up to 33 % off (found by cutting config 5 into slabs: the cuts change which particles share a wave).  Now x
    travels as a coarse + a fine fp16 half (exact coarse differences), and a wave that is far apart in y or z is walked
    in passes, one reference point per group of lanes (the stragglers after three passes gather in fp32).  A long flat slab of fluid (100 x 3 x 5 lattice: every x-row is 31 h long and no row is a multiple of 64
    particles) plus a sprinkle of isolated particles: every density within the mixed tolerance of the oracle."""
    from gpufluidsimulator_amd import ic
    box, grid = (8.0, 8.0, 8.0), (128, 128, 128)
    pos, vel = ic.dam_break_lattice((100, 3, 5), box, jitter=True)
    rng = np.random.default_rng(5)
    spray = rng.uniform(-3.9, 3.9, (300, 3)).astype(np.float32)
    spray[:, 1] = np.abs(spray[:, 1])                               # above the fluid
    pos = np.concatenate([pos, spray]); vel = np.zeros_like(pos)
    o = oracle.Oracle(pos, vel, box, grid, oracle.CELL_LINEAR)
    o.map_zindex(); o.sort(); o.construct_bgrid(); o.compute_densities()
    want = o.by_index("density")
    o.close()
    res = {}
    with capi.Context(pos.shape[0], box=box, grid=grid) as c:
        for mixed in (False, True):
            c.set_precision(mixed)
            c.upload(pos, vel)
            c.hash(); c.sort(); c.build_cells(); c.density()
            res[mixed] = c.download(want=("density",))["density"]
    assert np.abs(res[False] / want - 1).max() <= 1e-5
    rel = res[True] / want - 1
    assert np.abs(rel).max() <= RHO_MAX and np.sqrt(np.mean(rel ** 2)) <= RHO_RMS, (np.abs(rel).max(), np.sqrt(np.mean(rel ** 2)))
    assert np.abs(rel[-300:]).max() <= 1e-3                          # the spray: lone particles, far apart in y and z too
