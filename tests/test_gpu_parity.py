"""GPU (MI355X) parity tests: the HIP library, called through its C ABI (ctypes), against the
CPU oracle on the same seeded inputs and against the fixtures generated from the REFERENCE's own
CPU code (tests/golden, oracle/make_golden.py).

Bars: integer / index work (cell keys, sort order, cell table, collision counts) bit-exact;
floating point within the stated tolerance below.  The GPU sums the same 27-cell candidate set
as the reference but in a different order (rows dz,dy outer; the reference dx,dy,dz), with FMA
contraction, x*x*x instead of powf(x,3) and v_rsq_f32 instead of sqrt+divide, so results differ
from the reference at the few-ulp level per pair.
"""
import numpy as np
import pytest

from conftest import load_golden
from gpufluidsimulator_amd import capi, ic
from oracle import oracle

pytestmark = pytest.mark.gpu

# ---- stated fp32 tolerance (north star: "within a stated fp32 tolerance on positions/velocities") ----
REL_TOL = 1e-5               # density, velocity (relative to the largest magnitude in the array)
# The collision term is discontinuous (a pair counts or not: d <= 2R and r.v < 0, and the count divides the whole
# delta-v).  From IDENTICAL inputs the GPU decides every pair like the reference (both predicates are evaluated in
# the reference's operation order: counts bit-exact in test_phases_vs_oracle and in the developed-flow lockstep
# tests, velocity within REL_TOL for EVERY particle).  In a FREE run the inputs of step k differ by ~1e-7 after
# k-1 steps, and a pair sitting on d = 2R or r.v = 0 can then be counted on one side only: measured on c1_jitter,
# the first such pair appears at step 42 and at step 100 one particle of 4096 carries count 2 instead of 3, i.e. a
# delta-v of 3.4e-4 instead of 2.55e-4 = 1.19e-5 of |v|max (test_free_run_outliers_are_collision_count_flips pins
# exactly that: every particle beyond REL_TOL has a flipped count).  The same C code built with FMA contraction
# shows the same flips on the CPU.  Free-run comparisons therefore allow a few particles
# beyond REL_TOL, none beyond OUTLIER_REL_TOL.
OUTLIER_FRACTION = 1e-3      # at most this share of the particles may exceed REL_TOL in velocity (free runs only) ...
OUTLIER_REL_TOL = 1e-4       # ... and none may exceed this
POS_TOL_PER_BOX = 1e-6       # position: absolute, times the box edge
FORCE_REL_TOL = 2e-5         # per-phase force arrays, relative to the largest force magnitude


def _ctx(g, capacity=None):
    n = g["pos"].shape[0]
    return capi.Context(capacity or n, box=g["box"], grid=g["grid"])


def _oracle_linear(g):
    return oracle.Oracle(g["pos"], g["vel"], g["box"], g["grid"], oracle.CELL_LINEAR)


def _assert_close(name, a, b, rel, scale=None):
    scale = float(np.abs(b).max()) if scale is None else scale
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    assert err <= rel * max(scale, 1e-30), f"{name}: max abs err {err:.3e} > {rel:g} * {scale:.3e}"
    return err / max(scale, 1e-30)


@pytest.mark.parametrize("name", ["c1_jitter", "random_clump"])
def test_hash_sort_cells_bit_exact(name):
    g = load_golden(name)
    gx, gy, gz = (int(v) for v in g["grid"])
    with _ctx(g) as c:
        c.upload(g["pos"], g["vel"])
        c.hash()
        o = _oracle_linear(g)
        o.map_zindex()
        off = np.uint32(c.cell_key(0, 0, 0))                      # 0 for a whole-domain context (no ghost layer)
        want = o.particles["zindex"] + off
        assert np.array_equal(c.keys(), want)
        c.sort()
        order = np.argsort(want, kind="stable").astype(np.uint32)
        assert np.array_equal(c.order(), order), "stable radix sort order"
        assert np.array_equal(c.keys(), want[order])
        c.build_cells()
        k, s, cnt = c.cells()
        o.sort(); o.construct_bgrid()
        B = o.B
        occ = np.nonzero(B["nParticles"])[0].astype(np.uint32)
        assert np.array_equal(k, occ + off)
        assert np.array_equal(s, B["start"][occ]) and np.array_equal(cnt, B["nParticles"][occ])
        o.close()


@pytest.mark.parametrize("name", ["c1_lattice", "c1_jitter", "random_clump", "c1_flow", "d24_flow"])
def test_phases_vs_oracle(name):
    """Each phase on the GPU against the oracle fed with the same particle order."""
    g = load_golden(name)
    dt = float(g["dt"])
    with _ctx(g) as c:
        c.upload(g["pos"], g["vel"])
        o = _oracle_linear(g)
        for step in range(2):
            c.hash(); c.sort(); c.build_cells()
            o.map_zindex(); o.sort(); o.apply_order(c.order()); o.construct_bgrid()
            c.density(); o.compute_densities()
            st = c.download(want=("density", "pressure"))
            _assert_close("density", st["density"], o.by_index("density"), REL_TOL)
            _assert_close("pressure", st["pressure"], o.by_index("pressure"), REL_TOL)
            c.force(); o.compute_forces()
            c.collide(); o.particle_collisions()
            f = c.download_forces()
            fscale = float(max(np.abs(o.by_index("force_press")).max(), np.abs(o.by_index("force_visc")).max()))
            _assert_close("f_press", f["fpress"], o.by_index("force_press"), FORCE_REL_TOL, fscale)
            _assert_close("f_visc", f["fvisc"], o.by_index("force_visc"), FORCE_REL_TOL, fscale)
            # integer output: the same pairs collide (both predicates of computeCollision are evaluated in the
            # reference's own operation order, see k_force's drain loop)
            assert np.array_equal(f["count"], o.by_index("collision_count")), "collision counts"
            dvs = float(np.abs(o.by_index("delta_velocity")).max())
            _assert_close("delta_v", f["dv"], o.by_index("delta_velocity"), FORCE_REL_TOL, max(dvs, 1e-12))
            c.integrate(dt); o.integrate(dt)
            s = c.download()
            so = o.state()
            assert np.abs(s["pos"] - so["pos"]).max() <= POS_TOL_PER_BOX * float(g["box"].max())
            _assert_close("velocity", s["vel"], so["vel"], REL_TOL)
            p4 = c.positions4()
            assert np.array_equal(p4[:, :3], s["pos"]) and np.all(p4[:, 3] == 1.0)
        o.close()


@pytest.mark.parametrize("name", ["c1_lattice", "c1_jitter"])
def test_c1_100_steps_vs_reference_golden(name):
    """BASELINE config 1 (4096 particles, 64^3 grid, dt 5e-7): states after 1, 10 and 100 steps
    against what the reference's own CPU code produced (tests/golden)."""
    g = load_golden(name)
    box = float(g["box"].max())
    with _ctx(g) as c:
        c.upload(g["pos"], g["vel"])
        done = 0
        for s in (1, 10, 100):
            c.step(float(g["dt"]), s - done)
            done = s
            st = c.download()
            ref = g[f"state_{s}"]
            assert np.abs(st["pos"] - ref[:, 0:3]).max() <= POS_TOL_PER_BOX * box, f"step {s} position"
            ev = np.abs(st["vel"] - ref[:, 3:6]).max(axis=1) / np.abs(ref[:, 3:6]).max()
            assert ev.max() <= OUTLIER_REL_TOL, f"step {s} velocity: worst particle {ev.max():.2e}"
            assert (ev > REL_TOL).mean() <= OUTLIER_FRACTION, f"step {s}: {(ev > REL_TOL).sum()} velocity outliers"
            assert np.median(ev) <= 1e-6
            assert np.abs(st["density"] / ref[:, 6] - 1).max() <= REL_TOL, f"step {s} density"


@pytest.mark.parametrize("name,fused", [("c1_flow", True), ("c1_flow", False), ("d24_flow", True)])
def test_developed_flow_lockstep_vs_reference_golden(name, fused):
    """Developed flow (tests/golden/*_flow.npz: the reference's dam after 4000 steps -- particles changing cell,
    wall hits, thousands of colliding pairs).  Lockstep: upload the reference's state k, step ONCE, compare with
    the reference's state k+1, so that every step is checked from identical inputs and chaos cannot accumulate.
    The full bar, no outlier clause: positions 1e-6 x box, velocity and density 1e-5."""
    g = load_golden(name)
    box, dt, n = float(g["box"].max()), float(g["dt"]), g["pos"].shape[0]
    pos, vel = g["pos"], g["vel"]
    with _ctx(g) as c:
        for k in range(1, int(g["lock_steps"]) + 1):
            c.upload(pos, vel)
            (c.step if fused else c.step_phased)(dt, 1)
            st, ref = c.download(), g[f"state_{k}"]
            assert np.abs(st["pos"] - ref[:, 0:3]).max() <= POS_TOL_PER_BOX * box, f"step {k} position"
            _assert_close(f"step {k} velocity", st["vel"], ref[:, 3:6], REL_TOL)
            assert np.abs(st["density"] / ref[:, 6] - 1).max() <= REL_TOL, f"step {k} density"
            _assert_close(f"step {k} pressure", st["pressure"], ref[:, 7], REL_TOL)
            pos, vel = np.ascontiguousarray(ref[:, 0:3]), np.ascontiguousarray(ref[:, 3:6])
    assert n == ref.shape[0]


@pytest.mark.parametrize("name", ["c1_flow", "d24_flow"])
def test_developed_flow_phase_records_vs_reference_golden(name):
    """The per-phase records of the reference's first step from the developed state: density/pressure, both force
    terms, delta_v and the collision COUNTS (bit-exact: integer output), through the phase API."""
    g = load_golden(name)
    with _ctx(g) as c:
        c.upload(g["pos"], g["vel"])
        c.hash(); c.sort(); c.build_cells(); c.density(); c.force(); c.collide()
        st, f = c.download(want=("density", "pressure")), c.download_forces()
        dens, force, coll = g["s1_dens"], g["s1_force"], g["s1_coll"]
        assert np.abs(st["density"] / dens[:, 0] - 1).max() <= REL_TOL
        _assert_close("pressure", st["pressure"], dens[:, 1], REL_TOL)
        fscale = float(np.abs(force).max())
        _assert_close("f_press", f["fpress"], force[:, 0:3], FORCE_REL_TOL, fscale)
        _assert_close("f_visc", f["fvisc"], force[:, 3:6], FORCE_REL_TOL, fscale)
        assert np.array_equal(f["count"], coll[:, 3].astype(np.int32)), "collision counts"
        assert int((coll[:, 3] > 0).sum()) > 0.5 * coll.shape[0]          # the fixture really is full of collisions
        _assert_close("delta_v", f["dv"], coll[:, 0:3], FORCE_REL_TOL, float(np.abs(coll[:, 0:3]).max()))


def test_free_run_outliers_are_collision_count_flips():
    """GPU and oracle both run FREELY for 100 steps from c1_jitter; the collision counts of every step are compared.
    Until the first count differs every particle is within REL_TOL; afterwards a particle may leave REL_TOL only
    if its own collision count (or that of a particle within collision range of it) has differed at some step."""
    g = load_golden("c1_jitter")
    dt, n = float(g["dt"]), g["pos"].shape[0]
    flipped = np.zeros(n, dtype=bool)
    with _ctx(g) as c:
        c.upload(g["pos"], g["vel"])
        o = _oracle_linear(g)
        for s in range(1, 101):
            c.hash(); c.sort(); c.build_cells(); c.density(); c.force(); c.collide()
            o.map_zindex(); o.sort(); o.apply_order(c.order()); o.construct_bgrid()
            o.compute_densities(); o.compute_forces(); o.particle_collisions()
            mism = c.download_forces(force=False)["count"] != o.by_index("collision_count")
            flipped |= mism
            c.integrate(dt); o.integrate(dt)
            if s % 10 == 0 or mism.any():
                st, so = c.download(want=("pos", "vel")), o.state()
                ev = np.abs(st["vel"] - so["vel"]).max(axis=1) / np.abs(so["vel"]).max()
                if not flipped.any():
                    assert ev.max() <= REL_TOL, f"step {s}: {ev.max():.2e} with identical collision histories"
                else:
                    # partners of a flipped particle feel it through their own delta-v one step later
                    near = np.zeros(n, dtype=bool)
                    for k in np.nonzero(flipped)[0]:
                        near |= np.linalg.norm(so["pos"] - so["pos"][k], axis=1) <= 4.0 / 64.0
                    bad = ev > REL_TOL
                    assert not (bad & ~near).any(), f"step {s}: a particle without a flipped pair is off by {ev[bad & ~near].max():.2e}"
                    assert ev.max() <= OUTLIER_REL_TOL and bad.mean() <= OUTLIER_FRACTION
        o.close()
    assert flipped.sum() <= 8, f"{flipped.sum()} particles with a flipped collision count in 100 steps"


def test_random_clump_vs_reference_golden():
    """Walls, collisions, >32 particles in one cell, empty cell 0: 4 steps against the reference.
    Every step is first checked in LOCKSTEP -- from the reference's own state k-1, i.e. identical inputs -- at the FULL
    bar: position 1e-6 x box, velocity and density 1e-5 for every particle, no outlier clause, collision counts
    bit-exact (step 1: the fixture's per-phase record).  The free run (no re-synchronisation) is held to the documented
    free-run clause: at most 0.1 % of the particles beyond 1e-5, none beyond 1e-4."""
    g = load_golden("random_clump")
    dt, box = float(g["dt"]), float(g["box"].max())
    with _ctx(g) as c:
        # lockstep, phase by phase for step 1
        c.upload(g["pos"], g["vel"])
        c.hash(); c.sort(); c.build_cells(); c.density(); c.force(); c.collide()
        f = c.download_forces()
        assert np.array_equal(f["count"], g["s1_coll"][:, 3].astype(np.int32)), "step 1 collision counts"
        _assert_close("step 1 delta_v", f["dv"], g["s1_coll"][:, 0:3], FORCE_REL_TOL)
        c.integrate(dt)
        pos, vel = g["pos"], g["vel"]
        for s in (1, 2, 3, 4):
            if s > 1:
                c.upload(pos, vel)
                c.step(dt, 1)
            st, ref = c.download(), g[f"s{s}_state"]
            assert np.abs(st["pos"] - ref[:, 0:3]).max() <= POS_TOL_PER_BOX * box, f"lockstep {s} position"
            _assert_close(f"lockstep {s} velocity", st["vel"], ref[:, 3:6], REL_TOL)
            assert np.abs(st["density"] / ref[:, 6] - 1).max() <= REL_TOL, f"lockstep {s} density"
            pos, vel = np.ascontiguousarray(ref[:, 0:3]), np.ascontiguousarray(ref[:, 3:6])
        # free run
        c.upload(g["pos"], g["vel"])
        for s in (1, 2, 3, 4):
            c.step_phased(dt, 1)
            st, ref = c.download(), g[f"s{s}_state"]
            assert np.abs(st["density"] / ref[:, 6] - 1).max() <= REL_TOL, f"step {s} density"
            ev = np.abs(st["vel"] - ref[:, 3:6]).max(axis=1) / np.abs(ref[:, 3:6]).max()
            assert ev.max() <= OUTLIER_REL_TOL, f"step {s} velocity: worst particle {ev.max():.2e}"
            assert (ev > REL_TOL).mean() <= OUTLIER_FRACTION, f"step {s}: {(ev > REL_TOL).sum()} velocity outliers"
            ep = np.abs(st["pos"] - ref[:, 0:3]).max(axis=1)
            assert ep.max() <= 10 * POS_TOL_PER_BOX * box and (ep > POS_TOL_PER_BOX * box).mean() <= OUTLIER_FRACTION, f"step {s} position"


def test_c2_developed_flow_vs_reference_golden():
    """BASELINE config 2's geometry (262,144 particles, 128^3 grid) in DEVELOPED FLOW: the reference's dam after 2600
    steps (tests/golden/c2_flow.npz: the full state is the input; outputs are every 61st particle + checksums of the
    full arrays, collision counts of step 1 for EVERY particle).  Step 1 from identical inputs at the full bar, phase by
    phase (density, both forces, delta-v, counts bit-exact) and fused; step 2 is a two-step free run under the free-run
    clause."""
    g = load_golden("c2_flow")
    dt, box, sample = float(g["dt"]), float(g["box"].max()), g["sample"]
    n = g["pos"].shape[0]
    assert n == 262144 and int((g["s1_coll_count"] > 0).sum()) > 0.5 * n
    with capi.Context(n, box=g["box"], grid=g["grid"]) as c:
        c.upload(g["pos"], g["vel"])
        c.hash(); c.sort(); c.build_cells(); c.density(); c.force(); c.collide()
        st, f = c.download(want=("density", "pressure")), c.download_forces()
        assert np.array_equal(f["count"].astype(np.uint8), g["s1_coll_count"]) and f["count"].max() < 256, "collision counts"
        dens, force, coll = g["s1_dens_sample"], g["s1_force_sample"], g["s1_coll_sample"]
        assert np.abs(st["density"][sample] / dens[:, 0] - 1).max() <= REL_TOL
        _assert_close("pressure", st["pressure"][sample], dens[:, 1], REL_TOL)
        fscale = float(np.abs(force).max())
        _assert_close("f_press", f["fpress"][sample], force[:, 0:3], FORCE_REL_TOL, fscale)
        _assert_close("f_visc", f["fvisc"][sample], force[:, 3:6], FORCE_REL_TOL, fscale)
        _assert_close("delta_v", f["dv"][sample], coll[:, 0:3], FORCE_REL_TOL, float(np.abs(coll[:, 0:3]).max()))
        # checksums of the FULL arrays (sum of absolute values, float64)
        full = np.concatenate([st["density"][:, None], st["pressure"][:, None]], axis=1).astype(np.float64)
        assert np.all(np.abs(np.abs(full).sum(axis=0) / g["s1_dens_abs_sum"] - 1) <= REL_TOL)
        full = np.concatenate([f["fpress"], f["fvisc"]], axis=1).astype(np.float64)
        assert np.all(np.abs(np.abs(full).sum(axis=0) / g["s1_force_abs_sum"] - 1) <= FORCE_REL_TOL)
        c.integrate(dt)
        for fused in (False, True):
            if fused:
                c.upload(g["pos"], g["vel"])
                c.step(dt, 1)
            st, ref = c.download(), g["state_1_sample"]
            assert np.abs(st["pos"][sample] - ref[:, 0:3]).max() <= POS_TOL_PER_BOX * box
            _assert_close("step 1 velocity", st["vel"][sample], ref[:, 3:6], REL_TOL)
            assert np.abs(st["density"][sample] / ref[:, 6] - 1).max() <= REL_TOL
            full = np.concatenate([st["pos"], st["vel"], st["density"][:, None], st["pressure"][:, None]], axis=1)
            assert np.all(np.abs(np.abs(full.astype(np.float64)).sum(axis=0) / g["state_1_abs_sum"] - 1) <= REL_TOL)
        c.step(dt, 1)                                   # step 2: free
        st, ref = c.download(), g["state_2_sample"]
        assert np.abs(st["pos"][sample] - ref[:, 0:3]).max() <= POS_TOL_PER_BOX * box
        ev = np.abs(st["vel"][sample] - ref[:, 3:6]).max(axis=1) / np.abs(ref[:, 3:6]).max()
        assert ev.max() <= OUTLIER_REL_TOL and (ev > REL_TOL).mean() <= OUTLIER_FRACTION
        assert np.abs(st["density"][sample] / ref[:, 6] - 1).max() <= REL_TOL
        stats = c.sort_stats()
        assert stats["movers_total"] > 0                # particles changed cell inside the stored steps


def test_fused_step_equals_phased_step():
    g = load_golden("c1_jitter")
    with _ctx(g) as a, _ctx(g) as b:
        a.upload(g["pos"], g["vel"]); b.upload(g["pos"], g["vel"])
        a.step(float(g["dt"]), 5)
        b.step_phased(float(g["dt"]), 5)
        sa, sb = a.download(), b.download()
        assert np.abs(sa["pos"] - sb["pos"]).max() <= 1e-7 * 4
        _assert_close("velocity", sa["vel"], sb["vel"], 2e-6)
        assert np.array_equal(sa["density"], sb["density"])


def test_c2_sample_vs_reference_golden():
    """BASELINE config 2: 262144 particles (64^3 lattice), 128^3 grid, states after 1, 2, 3 and 30
    steps; every 61st particle plus checksums of the full arrays from the reference run."""
    g = load_golden("c2_sample")
    lattice = tuple(int(v) for v in g["lattice"])
    pos, vel = ic.dam_break_lattice(lattice, g["box"], jitter=True)
    sample = g["sample"]
    with capi.Context(pos.shape[0], box=g["box"], grid=g["grid"]) as c:
        c.upload(pos, vel)
        done = 0
        for s in (1, 2, 3, 30):
            c.step(float(g["dt"]), s - done)
            done = s
            st = c.download()
            ref = g[f"state_{s}_sample"]
            assert np.abs(st["pos"][sample] - ref[:, 0:3]).max() <= POS_TOL_PER_BOX * 8.0
            ev = np.abs(st["vel"][sample] - ref[:, 3:6]).max(axis=1) / np.abs(ref[:, 3:6]).max()
            assert ev.max() <= OUTLIER_REL_TOL and (ev > REL_TOL).mean() <= OUTLIER_FRACTION, f"step {s} velocity"
            assert np.abs(st["density"][sample] / ref[:, 6] - 1).max() <= REL_TOL
            full = np.concatenate([st["pos"], st["vel"], st["density"][:, None], st["pressure"][:, None]], axis=1)
            got_abs = np.abs(full.astype(np.float64)).sum(axis=0)
            assert np.all(np.abs(got_abs / g[f"state_{s}_abs_sum"] - 1) <= REL_TOL), f"step {s} checksums"


def test_ragged_and_empty_inputs():
    box, grid = (2.0, 2.0, 2.0), (32, 32, 32)
    with capi.Context(1000, box=box, grid=grid) as c:
        c.upload(np.zeros((0, 3), np.float32))           # empty
        c.step(1e-6, 2)
        assert c.n == 0
        pos, vel = ic.random_box(777, box, speed=5.0, fill=0.3)   # not a multiple of 64
        c.upload(pos, vel)
        c.step(1e-6, 3)
        o = oracle.Oracle(pos, vel, box, grid, oracle.CELL_LINEAR)
        o.step(1e-6, 3)
        st, so = c.download(count=777), o.state()
        assert np.abs(st["pos"] - so["pos"]).max() <= POS_TOL_PER_BOX * 2.0
        assert np.abs(st["density"] / so["density"] - 1).max() <= REL_TOL
        one = np.float32([[0.1, 0.2, 0.3]])              # a single particle: self density only
        c.upload(one)
        c.hash(); c.sort(); c.build_cells(); c.density()
        rho = c.download(count=1)["density"][0]
        assert abs(rho / (65.0 * 315.0 / (65.0 * np.pi * 0.1 ** 9) * 1e-6) - 1) < 1e-5
        o.close()


def test_errors_are_reported():
    with capi.Context(64, box=(2, 2, 2), grid=(32, 32, 32)) as c:
        with pytest.raises(capi.SphError):
            c.upload(np.zeros((65, 3), np.float32))      # over capacity
        with pytest.raises(capi.SphError):
            c.density()                                  # phase out of order
    with pytest.raises(capi.SphError):
        capi.Context(64, box=(2, 2, 2), grid=(32, 32, 32), device=99)
