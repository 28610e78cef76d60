"""GPU: the sort's merge path (only the particles whose cell changed are sorted, sph_sort.hip:
launch_sort_merge) must produce the SAME (key, slot) sequence as the full stable radix sort -- element
for element, for any number of movers -- so every later array is bit-identical too.

Two contexts run the same input, one created with SPH_SORT_MERGE=0 (always the full sort), one with the
merge path; keys, permutation and state are compared exactly after every step."""
import os

import numpy as np
import pytest
import torch  # noqa: F401  -- before libsph_hip.so is loaded: one HIP runtime per process (capi.load)

from gpufluidsimulator_amd import capi, ic

pytestmark = pytest.mark.gpu


def _ctx(n, box, grid, merge):
    old = os.environ.get("SPH_SORT_MERGE")
    os.environ["SPH_SORT_MERGE"] = "1" if merge else "0"        # read once, by sph_create
    try:
        return capi.Context(n, box=box, grid=grid)
    finally:
        if old is None:
            del os.environ["SPH_SORT_MERGE"]
        else:
            os.environ["SPH_SORT_MERGE"] = old


def _lockstep(pos, vel, box, grid, dt, steps, fused, full_table=False, always=False):
    n = pos.shape[0]
    a, b = _ctx(n, box, grid, False), _ctx(n, box, grid, True)
    if always:
        b.set_sort_mode(2)                 # the merge path whatever the mover count
    movers = []
    try:
        for c in (a, b):
            c.upload(pos, vel)
        for s in range(steps):
            for c in (a, b):
                if fused:
                    c.step(dt, 1)
                else:
                    c.step_phased(dt, 1)
            ka, kb = a.keys(), b.keys()
            assert np.array_equal(ka, kb), f"step {s}: sorted keys differ"
            assert np.all(np.diff(ka.astype(np.int64)) >= 0)
            assert np.array_equal(a.order(), b.order()), f"step {s}: permutation differs"
            movers.append(b.sort_stats()["last_movers"])
        if full_table:                     # every cell of the table, the empty ones included
            ncells = int(np.prod(grid))
            ta = [a.cell_range(k) for k in range(ncells)]
            assert ta == [b.cell_range(k) for k in range(ncells)]
            occupied = set(int(k) for k in ka)
            assert all((r == (0, 0)) == (k not in occupied) for k, r in enumerate(ta))
        sa, sb = a.download(), b.download()
        for k in ("pos", "vel", "density", "pressure"):
            assert np.array_equal(sa[k], sb[k], equal_nan=True), k
        st_a, st_b = a.sort_stats(), b.sort_stats()
        assert st_a["merges"] == 0 and st_a["sorts"] == steps
        assert st_b["sorts"] == steps
        return st_b, movers
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("fused", [True, False])
def test_few_movers_dam_break(fused):
    cfg = ic.CONFIGS["C1"]
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=True)
    rng = np.random.default_rng(5)
    vel = rng.uniform(-60, 60, pos.shape).astype(np.float32)       # a few particles cross a cell face per step
    st, movers = _lockstep(pos, vel, cfg["box"], cfg["grid"], 4e-5, 16, fused)
    assert st["merges"] == 15                                       # every sort but the first
    assert 0 < max(movers) < pos.shape[0] // 8


def test_many_movers_and_the_fallback():
    """Half of the particles change cell every step: the merge is still exact (its grids loop) when it is forced;
    left to itself the sort falls back to the full radix sort as soon as the device has reported such a count
    (the integrate epilogue of every step counts the movers of the next sort)."""
    box, grid = (2.0, 2.0, 2.0), (32, 32, 32)
    pos, vel = ic.random_box(20000, box, speed=60.0, fill=0.45)
    st, movers = _lockstep(pos, vel, box, grid, 5e-4, 20, True, always=True)
    assert max(movers) > pos.shape[0] // 8
    assert st["merges"] == 19
    st, movers = _lockstep(pos, vel, box, grid, 5e-4, 20, True)
    assert max(movers) > pos.shape[0] // 8
    assert st["merges"] < 19


def test_clump_and_empty_cells():
    """Movers into cells that were empty, out of cells that become empty, and a cell holding 150 particles."""
    box, grid, n = (2.0, 2.0, 2.0), (32, 32, 32), 6000
    pos, vel = ic.random_box(n, box, speed=40.0, fill=0.45)
    rng = np.random.default_rng(7)
    pos[:150] = (np.float32([-0.5, -0.5, -0.5]) + rng.uniform(0.002, 0.060, (150, 3))).astype(np.float32)
    vel[:150] = rng.uniform(-300, 300, (150, 3)).astype(np.float32)      # the clump bursts
    st, movers = _lockstep(pos, vel, box, grid, 2e-5, 25, True, full_table=True)
    assert st["merges"] >= 20 and max(movers) > 0


def test_no_movers_at_all():
    """A lattice at rest: no particle changes cell, so from the second sort on (the first is the full sort; the
    integrate epilogue of every step counts the movers of the next sort) the merge path finds nothing to do and
    leaves order, keys and cell table as they are -- and they still equal what the full sort produces every step.
    (_lockstep reads the keys back after every step, i.e. the host is in lockstep with the device: the count is
    only looked at when it is already there, the host never waits for it.)"""
    box, grid = (2.0, 2.0, 2.0), (32, 32, 32)
    pos, vel = ic.dam_break_lattice((8, 8, 8), box, jitter=False)
    st, movers = _lockstep(pos, vel, box, grid, 5e-7, 8, True, full_table=True)
    assert movers == [0] * 8
    assert st["merges"] == 7 and st["skips"] == 7


def test_queued_steps_at_rest_never_wait_and_stay_exact():
    """A host that queues many steps at once runs ahead of the device: the mover count is not there yet when a
    sort is issued, so nothing is skipped (and nothing is waited for) -- the merge path does the work for 0 movers."""
    box, grid = (2.0, 2.0, 2.0), (32, 32, 32)
    pos, vel = ic.dam_break_lattice((8, 8, 8), box, jitter=False)
    a, b = _ctx(pos.shape[0], box, grid, False), _ctx(pos.shape[0], box, grid, True)
    try:
        for c in (a, b):
            c.upload(pos, vel)
            c.step(5e-7, 12)
        st = b.sort_stats()
        assert st["merges"] == 11 and st["movers_total"] == 0
        assert np.array_equal(a.keys(), b.keys()) and np.array_equal(a.order(), b.order())
        sa, sb = a.download(), b.download()
        for k in ("pos", "vel", "density", "pressure"):
            assert np.array_equal(sa[k], sb[k], equal_nan=True), k
    finally:
        a.close(); b.close()


def test_rest_then_motion():
    """Skipping must stop the moment something moves: at rest for a few steps, then a kick through the API."""
    box, grid = (2.0, 2.0, 2.0), (32, 32, 32)
    pos, vel = ic.dam_break_lattice((10, 10, 10), box, jitter=True)
    n = pos.shape[0]
    rng = np.random.default_rng(3)
    kick = rng.uniform(-200, 200, (n, 3)).astype(np.float32)
    a, b = _ctx(n, box, grid, False), _ctx(n, box, grid, True)
    try:
        for c in (a, b):
            c.upload(pos, vel)
            for _ in range(5):                    # a caller in lockstep with the device (one update per frame):
                c.step(5e-7, 1); c.sync()         # the mover count of the last step is there when the next sort starts
        assert b.sort_stats()["skips"] >= 3
        for c in (a, b):
            c.set_by_index(0, vel=kick)
            c.step(1e-4, 6)                       # now hundreds of particles change cell per step
        assert b.sort_stats()["last_movers"] > 0
        for c in (a, b):
            c.step(5e-7, 4)
        assert np.array_equal(a.keys(), b.keys()) and np.array_equal(a.order(), b.order())
        sa, sb = a.download(), b.download()
        for k in ("pos", "vel", "density", "pressure"):
            assert np.array_equal(sa[k], sb[k], equal_nan=True), k
    finally:
        a.close(); b.close()


def test_c2_size_fused_matches_full_sort():
    cfg = ic.CONFIGS["C2"]
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=True)
    rng = np.random.default_rng(11)
    vel = rng.uniform(-60, 60, pos.shape).astype(np.float32)
    st, movers = _lockstep(pos, vel, cfg["box"], cfg["grid"], 4e-5, 10, True)
    assert st["merges"] == 9 and max(movers) > 100


def test_mode_transitions_over_a_long_run():
    """Calm -> violent -> calm: the sort goes merge -> full (hint above N/8) -> merge again (the integrate
    epilogue keeps reporting the mover count while the full sort runs; steps are queued many at a time, the
    host stays at most four sorts ahead of the device), and the state stays bit-identical to the always-full-sort run."""
    cfg = ic.CONFIGS["C2"]
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=True)
    rng = np.random.default_rng(23)
    vel = rng.uniform(-40, 40, pos.shape).astype(np.float32)
    n = pos.shape[0]
    a, b = _ctx(n, cfg["box"], cfg["grid"], False), _ctx(n, cfg["box"], cfg["grid"], True)
    try:
        for c in (a, b):
            c.upload(pos, vel)
        merges = []
        for dt, steps in ((5e-6, 60), (4e-4, 24), (5e-6, 60), (1e-4, 30), (2e-6, 40)):
            for c in (a, b):
                c.step(dt, steps)                      # queued back to back: no sync, no fresh hint in between
            merges.append(b.sort_stats()["merges"])
            assert np.array_equal(a.keys(), b.keys())
            assert np.array_equal(a.order(), b.order())
        sa, sb = a.download(), b.download()
        for k in ("pos", "vel", "density", "pressure"):
            assert np.array_equal(sa[k], sb[k], equal_nan=True), k
        per_phase = np.diff([0] + merges)
        assert per_phase[0] >= 58                      # calm: every sort but the first merges
        assert per_phase[2] >= 50 and per_phase[4] >= 30   # back to merging within a few steps of the calm
        assert a.sort_stats()["merges"] == 0
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("movers", [1, 63, 4095, 4096, 4097, 8191, 8192, 8193, 20000])
def test_mover_counts_around_the_one_block_sort(movers):
    """Up to 8192 movers the one-block sort (k_os_small) takes the movers and the generic kernels leave; from 8193 on it is
    the other way round.  The count lives on the device, so both forms are launched every time: exactly `movers` particles
    are pushed across a cell face (C2-size lattice, merge forced), and the order must be the full radix sort's, element for
    element, on either side of the switch."""
    cfg = ic.CONFIGS["C2"]
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=False)
    n = pos.shape[0]
    rng = np.random.default_rng(movers)
    pick = rng.choice(n, size=movers, replace=False)
    a, b = _ctx(n, cfg["box"], cfg["grid"], False), _ctx(n, cfg["box"], cfg["grid"], True)
    b.set_sort_mode(2)
    try:
        for c in (a, b):
            c.upload(pos, vel)
            c.hash(); c.sort()                         # the order of the lattice
        moved = pos.copy()
        moved[pick, 0] += np.float32(1.0 / 16.0)       # one cell to the right (the lattice fills a quarter of the box)
        # positions by creation index: rewrite all of them in one call
        for c in (a, b):
            c.set_by_index(0, pos=moved)
            c.hash(); c.sort()
        assert np.array_equal(a.keys(), b.keys())
        assert np.array_equal(a.order(), b.order())
        st = b.sort_stats()
        assert st["last_movers"] == movers and st["merges"] == 1
        c2 = [c.build_cells() or c.cells(max_cells=n) for c in (a, b)]
        for x, y in zip(c2[0], c2[1]):
            assert np.array_equal(x, y)
    finally:
        a.close(); b.close()


def test_a_whole_domain_context_guesses_the_sort_form_and_a_wrong_guess_is_still_exact():
    """Round 5: a whole-domain context, too, launches only the form of the movers' sort that the last reported count asks for
    (its host runs at most four sorts ahead of the device).  Each form is exact for any count on its own: 5 movers (the
    one-block sort alone), then 20000 at once (the guess says "few": the one-block sort works through them tile by tile),
    then 5 again (the guess says "many": the multi-block passes sort five pairs) -- the order of the full radix sort every
    time, and the counters show which forms ran."""
    cfg = ic.CONFIGS["C2"]
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=False)
    n = pos.shape[0]
    rng = np.random.default_rng(11)
    a, b = _ctx(n, cfg["box"], cfg["grid"], False), _ctx(n, cfg["box"], cfg["grid"], True)
    b.set_sort_mode(2)
    try:
        for c in (a, b):
            c.upload(pos, vel)
            c.hash(); c.sort()
        cur = pos.copy()
        forms = [b.sort_forms()]
        for movers in (5, 5, 20000, 5, 5):
            pick = rng.choice(n, size=movers, replace=False)
            step = np.where(cur[pick, 0] < 0.0, 1.0, -1.0).astype(np.float32) / np.float32(16.0)      # one cell, staying in the box
            cur[pick, 0] += step
            for c in (a, b):
                c.set_by_index(0, pos=cur)
                c.trust_mover_hint()       # (a caller that rewrites particles gets both forms for five sorts: here the FLOW is meant to have moved them)
                c.hash(); c.sort()
            assert np.array_equal(a.keys(), b.keys()) and np.array_equal(a.order(), b.order()), movers
            assert b.sort_stats()["last_movers"] == movers
            forms.append(b.sort_forms())
        d = [tuple(y - x for x, y in zip(f0, f1)) for f0, f1 in zip(forms, forms[1:])]      # (both, small alone, passes alone) per sort
        assert d[1] == (0, 1, 0), d          # count 5 known: the one-block sort alone
        assert d[2] == (0, 1, 0), d          # 20000 movers behind a count of 5: still the one-block sort, alone and exact
        assert d[3] == (0, 0, 1), d          # 5 movers behind a count of 20000: the passes alone
        assert d[4] == (0, 1, 0), d
        # without the hook: new particle data from the caller -> both forms for the next five sorts (the count the device
        # last reported says nothing about them: a kick of every particle must not go through the one-block fallback)
        f0 = b.sort_forms()
        pick = rng.choice(n, size=30000, replace=False)
        cur[pick, 0] += np.where(cur[pick, 0] < 0.0, 1.0, -1.0).astype(np.float32) / np.float32(16.0)
        for c in (a, b):
            c.set_by_index(0, pos=cur)
            c.hash(); c.sort()
        assert np.array_equal(a.keys(), b.keys()) and np.array_equal(a.order(), b.order())
        assert tuple(y - x for x, y in zip(f0, b.sort_forms())) == (1, 0, 0)
    finally:
        a.close(); b.close()
