"""GPU: the two ways the neighbour passes reach a particle's candidates give the same bits.

A wave stages, per (dz, dy) row, the hull of its lanes' candidate ranges through LDS -- unless one of its hulls is
longer than `direct_hull` slots (sparse particles next to a dense layer: csrc/sph_pairs.hip, traverse), in which case
every lane reads its own candidates from global memory.  Which way a particle's wave goes depends on the other 63
particles of the wave, i.e. on how the domain is cut into slabs, so the two ways must agree bit for bit: same
candidates, same order, same arithmetic."""
import numpy as np
import pytest

from conftest import bits, load_golden
from gpufluidsimulator_amd import capi, ic

pytestmark = pytest.mark.gpu
DT = 5e-7
NEVER = 0xFFFFFFFF


def _phases(pos, vel, box, grid, direct_hull, mixed=False, small_blocks=None):
    with capi.Context(pos.shape[0], box=box, grid=grid) as c:
        c.set_direct_hull(direct_hull)
        if small_blocks is not None:
            c.set_pair_small_launch(NEVER if small_blocks else 0)
        c.set_precision(mixed)
        c.upload(pos, vel)
        c.hash(); c.sort(); c.build_cells(); c.density(); c.force(); c.collide()
        out = dict(c.download(), **c.download_forces())
        c.step(DT, 4)
        out.update({k + "_4": v for k, v in c.download().items()})
    return out


def _sparse_next_to_dense():
    """A dense 24 x 24 x 12 block and, one cell layer above it, 150 particles scattered over the whole layer: 64
    consecutive ones of those span many y-rows, their dz = -1 rows cover whole rows of the dense block."""
    box, grid = (4.0, 4.0, 4.0), (64, 64, 64)
    pos, vel = ic.dam_break_lattice((24, 24, 12), box, jitter=True)
    rng = np.random.default_rng(7)
    top = pos[:, 2].max()
    cell = box[2] / grid[2]
    z_layer = (np.floor((top + box[2] / 2) / cell) + 1.3) * cell - box[2] / 2       # inside the next cell layer
    extra = np.empty((150, 3), np.float32)
    extra[:, 0] = rng.uniform(pos[:, 0].min(), pos[:, 0].max(), 150)
    extra[:, 1] = rng.uniform(pos[:, 1].min(), pos[:, 1].max(), 150)
    extra[:, 2] = z_layer + rng.uniform(-0.01, 0.01, 150)
    pos = np.concatenate([pos, extra]).astype(np.float32)
    vel = np.concatenate([vel, np.zeros((150, 3), np.float32)])
    vel[:, 2] = rng.uniform(-3000, 3000, pos.shape[0]).astype(np.float32)
    return pos, vel, box, grid


@pytest.mark.parametrize("case", ["sparse_next_to_dense", "c1_flow"])
def test_direct_and_staged_rows_agree_bit_for_bit(case):
    if case == "c1_flow":
        g = load_golden("c1_flow")
        pos, vel = g["pos"], g["vel"]
        box, grid = g["box"], g["grid"]
    else:
        pos, vel, box, grid = _sparse_next_to_dense()
    ref = _phases(pos, vel, box, grid, NEVER)
    for hull in (0, 100, 512):
        got = _phases(pos, vel, box, grid, hull)
        for k in ref:
            assert np.array_equal(bits(got[k]), bits(ref[k])) if ref[k].dtype == np.float32 else np.array_equal(got[k], ref[k]), (case, hull, k)


@pytest.mark.parametrize("case", ["sparse_next_to_dense", "c1_flow"])
def test_workgroups_of_128_and_of_256_threads_agree_bit_for_bit(case):
    """A context of fewer than 524,288 particles launches the neighbour passes in workgroups of 128 threads, larger ones in
    256 (sph_set_pair_small_launch): a wave's work does not depend on its block, so every phase output and four whole steps
    come out the same bits either way -- under the staged and under the direct walk."""
    if case == "c1_flow":
        g = load_golden("c1_flow")
        pos, vel, box, grid = g["pos"], g["vel"], g["box"], g["grid"]
    else:
        pos, vel, box, grid = _sparse_next_to_dense()
    for hull in (512, 0):
        ref = _phases(pos, vel, box, grid, hull, small_blocks=False)
        got = _phases(pos, vel, box, grid, hull, small_blocks=True)
        for k in ref:
            assert np.array_equal(bits(got[k]), bits(ref[k])) if ref[k].dtype == np.float32 else np.array_equal(got[k], ref[k]), (case, hull, k)


def test_direct_rows_in_mixed_precision_stay_within_the_mixed_tolerance():
    """The packed-fp16 density pass pairs a lane's candidates per staged piece, so its sums depend on the staging anyway
    (mixed): the direct walk is held to the mode's tolerance against the fp32 pass, not to bits."""
    pos, vel, box, grid = _sparse_next_to_dense()
    ref = _phases(pos, vel, box, grid, NEVER)
    for hull in (0, 512):
        got = _phases(pos, vel, box, grid, hull, mixed=True)
        err = np.abs(got["density"] / ref["density"] - 1)
        assert err.max() <= 0.02 and np.sqrt((err ** 2).mean()) <= 0.004, (hull, err.max())


def test_block_orders_agree_bit_for_bit():
    """The ORDER in which the pair kernels' workgroups take the slots (sph_set_block_order: plain, a contiguous eighth per
    XCD, strips through the cell layers of that eighth) changes memory traffic, not results: 2,097,152 particles -- enough
    workgroups and cell layers for the strips to engage -- stepped under four orders, and under two of them in workgroups of
    128 threads as well (sph_set_pair_small_launch: twice the blocks per cell layer), bit for bit."""
    box, grid = (16.0, 16.0, 16.0), (256, 256, 256)
    lattice = (128, 128, 128)
    n = lattice[0] * lattice[1] * lattice[2]
    ref = None
    for order in ((0, 0, 4), (1, 0, 4), (1, 1, 2), (1, 1, 4), (1, 1, 4, "small"), (0, 0, 4, "small")):
        with capi.Context(n, box=box, grid=grid) as c:
            c.set_block_order(*order[:3])
            c.set_pair_small_launch(NEVER if len(order) > 3 else 0)
            c.reset_lattice(lattice, jitter=True)
            c.step(DT, 6)
            got = c.download()
        if ref is None:
            ref = got
            continue
        for k in ref:
            assert np.array_equal(bits(got[k]), bits(ref[k])), (order, k)
