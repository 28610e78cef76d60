"""GPU: inputs off the beaten path -- grids that are not powers of two and not cubic (the generic key
decode), boxes with three different edges, fewer particles than one wave, everything in ONE cell
(candidate hulls far longer than one staged piece), shuffled creation indices, a box change through
sph_set_params, and particles faster than one cell per step."""
import numpy as np
import pytest
import torch  # noqa: F401  -- before libsph_hip.so is loaded: one HIP runtime per process (capi.load)

from gpufluidsimulator_amd import capi, ic
from oracle import oracle

pytestmark = pytest.mark.gpu
REL = 1e-5


def _compare(pos, vel, box, grid, dt, steps, index=None, rel=REL):
    n = pos.shape[0]
    with capi.Context(max(n, 1), box=box, grid=grid) as c:
        c.upload(pos, vel, index)
        c.step(dt, steps)
        st = c.download(count=n)
    o = oracle.Oracle(pos, vel, box, grid, oracle.CELL_LINEAR)
    o.step(dt, steps)
    so = o.state()
    o.close()
    if index is not None:                      # the context reports by the caller's creation index
        inv = np.empty(n, dtype=np.int64); inv[index] = np.arange(n)
        so = {k: v[inv] for k, v in so.items()}
    assert np.abs(st["pos"] - so["pos"]).max() <= 1e-6 * max(box)
    vs = max(np.abs(so["vel"]).max(), 1e-30)
    bad = np.abs(st["vel"] - so["vel"]).max(axis=1) > rel * vs
    assert bad.mean() <= 5e-3 and np.abs(st["vel"] - so["vel"]).max() <= 10 * rel * vs
    assert np.abs(st["density"] / so["density"] - 1).max() <= rel
    return st


def test_non_power_of_two_non_cubic_grid():
    box, grid = (3.0, 2.5, 3.5), (48, 40, 56)          # cell edge 0.0625 on every axis, nothing a power of two
    pos, vel = ic.dam_break_lattice((20, 14, 24), box, jitter=True)
    _compare(pos, vel, box, grid, 5e-7, 6)


def test_anisotropic_cells():
    box, grid = (2.0, 4.0, 8.0), (32, 32, 32)          # cells 0.0625 x 0.125 x 0.25
    pos, vel = ic.random_box(3000, box, speed=10.0, fill=0.35)
    _compare(pos, vel, box, grid, 1e-6, 4)


@pytest.mark.parametrize("n", [1, 2, 63, 65, 130])
def test_fewer_particles_than_a_wave(n):
    pos, vel = ic.random_box(n, (2.0, 2.0, 2.0), speed=3.0, fill=0.2)
    _compare(pos, vel, (2.0, 2.0, 2.0), (32, 32, 32), 1e-6, 3)


def test_everything_in_one_cell():
    """600 particles in a single cell: every lane's range is longer than a 128-entry piece."""
    rng = np.random.default_rng(3)
    pos = (np.float32([0.1, -0.4, 0.3]) + rng.uniform(0.001, 0.061, (600, 3))).astype(np.float32)
    pos = (np.floor((pos + 1.0) / 0.0625)[0] * 0.0625 - 1.0 + rng.uniform(0.002, 0.060, (600, 3))).astype(np.float32)
    vel = rng.uniform(-5, 5, (600, 3)).astype(np.float32)
    with capi.Context(600, box=(2.0,) * 3, grid=(32,) * 3) as c:
        c.upload(pos, vel)
        c.hash(); c.sort(); c.build_cells()
        k, s, cnt = c.cells()
        assert len(k) == 1 and cnt[0] == 600
    _compare(pos, vel, (2.0,) * 3, (32,) * 3, 2e-7, 2, rel=4e-5)


def test_shuffled_creation_indices():
    pos, vel = ic.dam_break_lattice((12, 12, 12), (2.0, 2.0, 2.0), jitter=True)
    rng = np.random.default_rng(11)
    index = rng.permutation(pos.shape[0]).astype(np.uint32)
    _compare(pos, vel, (2.0,) * 3, (32,) * 3, 5e-7, 5, index=index)


def test_set_params_moves_the_walls():
    """setSimParams analogue: shrink the box in y at run time; the clamp follows the new wall."""
    pos, vel = ic.dam_break_lattice((10, 10, 10), (2.0, 2.0, 2.0), jitter=True)
    with capi.Context(1000, box=(2.0,) * 3, grid=(32,) * 3) as c:
        c.upload(pos, vel)
        c.step(5e-7, 2)
        p = c.params
        p.box_min[1] = -0.9                               # floor up by 0.1: above the lowest lattice layers
        c.set_params(p)
        c.step(5e-7, 1)
        st = c.download(count=1000)
        assert st["pos"][:, 1].min() >= -0.9 + 0.9e-5
        q = capi.Params()
        capi._check(c.L.sph_get_params(c.h, q))
        assert abs(q.box_min[1] + 0.9) < 1e-7
        bad = capi.default_params((2.0,) * 3, (64,) * 3)
        with pytest.raises(capi.SphError):
            c.set_params(bad)                             # the grid cannot change


def test_faster_than_one_cell_per_step():
    pos, vel = ic.random_box(2000, (2.0, 2.0, 2.0), speed=0.0, fill=0.4)
    vel[:, 0] = 3.0e5                                     # 0.15 per step at dt 5e-7: more than two cells
    _compare(pos, vel, (2.0,) * 3, (32,) * 3, 5e-7, 3, rel=4e-5)


def _np_cell(p, bmin, bdim, g):
    q = ((p.astype(np.float32) - np.float32(bmin)) / np.float32(bdim)) * np.float32(g)
    return np.clip(np.floor(q).astype(np.int64), 0, int(g) - 1)


@pytest.mark.parametrize("grid,passes", [((1024, 1024, 512), "4x8"), ((512, 512, 512), "3x9"), ((128, 64, 32), "2x9"),
                                         ((64, 64, 16), "2x8")])
def test_every_radix_plan(grid, passes):
    """The sort picks 9-bit digits when they save a pass over 8-bit ones (29 key bits: 4 x 8, as on the
    huge table a last z-slab owns; 27: 3 x 9; 18: 2 x 9; 16: 2 x 8); hash, stable order and cell table
    are checked against numpy."""
    box = tuple(g / 16.0 for g in grid)
    n = 200_000
    pos, vel = ic.random_box(n, box, speed=0.0, fill=1.0)
    pos *= np.float32(0.999)
    with capi.Context(n, box=box, grid=grid) as c:
        c.upload(pos, vel)
        c.hash()
        cx, cy, cz = (_np_cell(pos[:, a], -box[a] / 2, box[a], grid[a]) for a in range(3))
        want = ((cz * grid[1] + cy) * grid[0] + cx).astype(np.uint32)
        assert np.array_equal(c.keys(), want)
        c.sort()
        order = np.argsort(want, kind="stable").astype(np.uint32)
        assert np.array_equal(c.order(), order) and np.array_equal(c.keys(), want[order])
        c.build_cells()
        k, s, cnt = c.cells()
        uk, ucnt = np.unique(want, return_counts=True)
        assert np.array_equal(k, uk) and np.array_equal(cnt, ucnt.astype(np.uint32))
        c.density()                                   # isolated particles: self term only
        rho = c.download(count=n, want=("density",))["density"]
        assert np.all(rho >= 0.999 * 315.0 / (np.pi * 1e-3))


def test_user_stream_and_two_contexts():
    """sph_set_stream with a caller-owned (torch) stream: same bits as the default stream; two contexts on
    two streams do not disturb each other."""
    import torch
    pos, vel = ic.dam_break_lattice((16, 16, 16), (4.0,) * 3, jitter=True)
    with capi.Context(4096, box=(4.0,) * 3, grid=(64,) * 3) as ref:
        ref.upload(pos, vel)
        ref.step(5e-7, 6)
        want = ref.download()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with capi.Context(4096, box=(4.0,) * 3, grid=(64,) * 3) as a, capi.Context(4096, box=(4.0,) * 3, grid=(64,) * 3) as b:
        a.set_stream(s1.cuda_stream); b.set_stream(s2.cuda_stream)
        a.upload(pos, vel); b.upload(pos[::-1].copy(), vel[::-1].copy(), np.arange(4095, -1, -1, dtype=np.uint32))
        for _ in range(6):
            a.step(5e-7, 1); b.step(5e-7, 1)
        ga, gb = a.download(), b.download()
    for k in ("pos", "vel", "density"):
        assert np.array_equal(ga[k].view(np.uint32), want[k].view(np.uint32)), k
    # b started from the reversed array: same physics, different summation order inside cells
    assert np.abs(gb["pos"] - want["pos"]).max() <= 1e-6 * 4.0
    assert np.abs(gb["density"] / want["density"] - 1).max() <= 1e-5


def test_set_by_index_equals_host_edit_and_reupload():
    """sph_set_by_index changes some particles in place on the device (by creation index, wherever the sort
    put them); the result must equal downloading the state, editing it on the host and uploading it again."""
    box, grid, n = (2.0, 2.0, 2.0), (32, 32, 32), 5000
    pos, vel = ic.random_box(n, box, speed=8.0, fill=0.4)
    rng = np.random.default_rng(41)
    first, count = 700, 900
    new_pos = rng.uniform(-0.6, 0.6, (count, 3)).astype(np.float32)
    new_vel = rng.uniform(-5, 5, (count, 3)).astype(np.float32)
    dt = 1e-6
    with capi.Context(n, box=box, grid=grid) as a:
        a.upload(pos, vel)
        a.step(dt, 3)
        mid = a.download()
        a.set_by_index(first, pos=new_pos, vel=new_vel)
        assert np.array_equal(a.positions4()[first:first + count, :3], new_pos)      # the by-index output follows
        with pytest.raises(capi.SphError):
            a.density()                                                                # stale: hash + sort come first
        a.step(dt, 3)
        sa = a.download()
        merges = a.sort_stats()["merges"]
    assert merges >= 2                                  # (900 of 5000 particles jump: the step after the edit sorts in full)
    p2, v2 = mid["pos"].copy(), mid["vel"].copy()
    p2[first:first + count] = new_pos
    v2[first:first + count] = new_vel
    with capi.Context(n, box=box, grid=grid) as b:
        b.upload(p2, v2)
        b.step(dt, 3)
        sb = b.download()
    assert np.abs(sa["pos"] - sb["pos"]).max() <= 1e-6 * max(box)
    assert np.abs(sa["vel"] - sb["vel"]).max() <= REL * np.abs(sb["vel"]).max()
    assert np.abs(sa["density"] / sb["density"] - 1).max() <= REL
    # position only / velocity only, and indices outside the range stay untouched
    with capi.Context(n, box=box, grid=grid) as c:
        c.upload(pos, vel)
        c.step(dt, 1)
        before = c.download()
        c.set_by_index(10, vel=new_vel[:5])
        c.set_by_index(4000, pos=new_pos[:7])
        c.hash(); c.sort()
        after = c.download(want=("pos", "vel"))
    exp_p, exp_v = before["pos"].copy(), before["vel"].copy()
    exp_v[10:15] = new_vel[:5]
    exp_p[4000:4007] = new_pos[:7]
    assert np.array_equal(after["pos"], exp_p) and np.array_equal(after["vel"], exp_v)
