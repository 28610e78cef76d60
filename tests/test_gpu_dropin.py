"""GPU: the drop-in claim, literally.  oracle/_ref/sph_ref_dropin is the REFERENCE's own host class
(SPH/particleSystem.cpp compiled unmodified, in the build container) linked against libsph_hip.so,
which exports the reference's 19 extern "C" seam symbols (include/sph_compat_seam.h).  The harness
calls the reference's own ParticleSystem::update() in CUDA_PARALLEL mode (particleSystem.cpp:769-801):
its cudaMapZIndex ... cudaIntegrate calls land in the HIP library.  The AoS it reads back must agree
with what the reference's CPU path produced (tests/golden)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import refio

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not refio.dropin_available(), reason="oracle/_ref/sph_ref_dropin not built")]


@pytest.mark.parametrize("name", ["c1_lattice", "c1_jitter"])
def test_reference_update_runs_on_the_hip_seam(name):
    g = load_golden(name)
    recs, stats = refio.run_ref(g["pos"], g["vel"], g["box"], int(g["grid"][0]), float(g["dt"]), 10,
                                dump_steps=(1, 10), binary=refio.DROPIN_BIN)
    for s in (1, 10):
        st, ref = recs[("state", s)], g[f"state_{s}"]
        assert np.abs(st[:, 0:3] - ref[:, 0:3]).max() <= 1e-6 * 4.0
        assert np.abs(st[:, 3:6] - ref[:, 3:6]).max() <= 1e-5 * np.abs(ref[:, 3:6]).max()
        assert np.abs(st[:, 6] / ref[:, 6] - 1).max() <= 1e-5
        assert np.abs(st[:, 7] - ref[:, 7]).max() <= 1e-5 * np.abs(ref[:, 7]).max()


def test_reference_update_random_clump():
    """Step 1 starts from identical inputs: the full bar (1e-6 x box, 1e-5, no outliers).  Step 2 is a free run of the
    reference's update(): the documented free-run clause (at most 0.1 % of the particles beyond 1e-5, none beyond 1e-4)."""
    g = load_golden("random_clump")
    recs, _ = refio.run_ref(g["pos"], g["vel"], g["box"], int(g["grid"][0]), float(g["dt"]), 2,
                            dump_steps=(1, 2), binary=refio.DROPIN_BIN)
    st, ref = recs[("state", 1)], g["s1_state"]
    assert np.abs(st[:, 0:3] - ref[:, 0:3]).max() <= 1e-6 * 2.0
    assert np.abs(st[:, 3:6] - ref[:, 3:6]).max() <= 1e-5 * np.abs(ref[:, 3:6]).max()
    assert np.abs(st[:, 6] / ref[:, 6] - 1).max() <= 1e-5
    st, ref = recs[("state", 2)], g["s2_state"]
    assert np.abs(st[:, 6] / ref[:, 6] - 1).max() <= 1e-5
    ev = np.abs(st[:, 3:6] - ref[:, 3:6]).max(axis=1) / np.abs(ref[:, 3:6]).max()
    assert ev.max() <= 1e-4 and (ev > 1e-5).mean() <= 1e-3
    ep = np.abs(st[:, 0:3] - ref[:, 0:3]).max(axis=1)
    assert ep.max() <= 1e-5 * 2.0 and (ep > 1e-6 * 2.0).mean() <= 1e-3


@pytest.mark.parametrize("name", ["c1_jitter", "random_clump"])
def test_seam_fills_B_and_Bprime_like_the_reference(name):
    """Bit-exact index work behind the reference's own interface.  The reference's host class runs two of its update()s on
    the HIP seam; what the seam left in the caller's arrays is compared `array_equal` -- no mapping between numberings --
    with what the reference's own code (its CPU path, tests/golden) leaves there:
      * Particle::zindex = coord2zIndex(cell), the reference's Morton code (particleSystem.cu:68-91, :307);
      * the particle array sorted by it (cudaSortParticles, :497-501);
      * dev_B[zindex] = {nParticles, start} for every occupied cell, zero elsewhere (:311-329, :503-509);
      * dev_B_prime = one {start, nParticles <= 32} entry per 32-particle chunk, and its size (:331-373, :511-528).
    (Only the order of the particles INSIDE a cell is unspecified in the reference -- std::sort / thrust::sort -- and is
    not compared.)"""
    g = load_golden(name)
    grid = int(g["grid"][0])
    recs, _ = refio.run_ref(g["pos"], g["vel"], g["box"], grid, float(g["dt"]), 2, phases=True, binary=refio.DROPIN_BIN)
    n = g["pos"].shape[0]
    for step in (1, 2):
        sorted_z, order = recs[("sorted_z", step)], recs[("order", step)]
        assert np.array_equal(sorted_z, g[f"s{step}_sorted_z"]), step
        assert np.array_equal(np.sort(order), np.arange(n, dtype=order.dtype))
        zindex = np.empty(n, np.uint32)
        zindex[order] = sorted_z                                   # by creation index
        ref_z = np.empty(n, np.uint32)
        ref_z[g[f"s{step}_order"]] = g[f"s{step}_sorted_z"]
        assert np.array_equal(zindex, ref_z), step
        if step == 1:                                              # the initial array is in creation order
            assert np.array_equal(zindex, g["s1_zindex"])
        assert np.array_equal(recs[("bcells", step)], g[f"s{step}_bcells"]), step      # {zindex, nParticles, start}
        assert np.array_equal(recs[("bprime", step)], g[f"s{step}_bprime"]), step      # {start, nParticles}


def test_morton_golden_through_the_seam():
    """cudaMapZIndex / cudaSortParticles / cudaConstructBGrid called directly (ctypes, device arrays from torch) on the
    `morton_c1` fixture: the z-indices the reference computed for these positions, the sorted order and a B table that
    describes exactly the runs of equal z-index."""
    import ctypes as C
    import torch
    from gpufluidsimulator_amd import capi
    g = load_golden("morton_c1")
    pos, want = g["pos"], g["zindex"]
    n, grid, box = pos.shape[0], int(g["grid"][0]), [float(b) for b in g["box"]]
    L = capi.load()
    aos = np.zeros((n, 22), np.uint32)
    aos[:, 0] = np.arange(n)
    aos[:, 1:4] = pos.view(np.uint32)
    aos[:, 16] = np.float32(65.0).view(np.uint32)
    aos[:, 19] = np.float32(1.0 / 64.0).view(np.uint32)
    prm = np.zeros(18, np.float32)
    prm[7] = 1.0 / 64.0
    prm[8:11] = [-b / 2 for b in box]; prm[11:14] = [b / 2 for b in box]; prm[14:17] = box
    prm.view(np.uint32)[17] = grid
    dev = torch.device("cuda", 0)
    d_aos = torch.from_numpy(aos.view(np.int32)).to(dev)
    d_prm = torch.from_numpy(prm).to(dev)
    b_size = grid ** 3
    d_B = torch.full((b_size, 2), -1, dtype=torch.int32, device=dev)
    vp = C.c_void_p
    L.cudaMapZIndex.argtypes = [vp, C.c_uint, vp]; L.cudaMapZIndex.restype = None
    L.cudaSortParticles.argtypes = [vp, C.c_uint]; L.cudaSortParticles.restype = None
    L.cudaConstructBGrid.argtypes = [vp, C.c_uint, vp, C.c_uint, vp]; L.cudaConstructBGrid.restype = None
    L.freeArray.argtypes = [vp]; L.threadSync.restype = None
    L.cudaMapZIndex(d_aos.data_ptr(), n, d_prm.data_ptr())
    L.threadSync()
    got = d_aos.cpu().numpy().view(np.uint32)
    assert np.array_equal(got[:, 21], want)                        # by array slot, before the sort
    assert np.array_equal(got[:, 0], np.arange(n))
    L.cudaSortParticles(d_aos.data_ptr(), n)
    L.cudaConstructBGrid(d_aos.data_ptr(), n, d_B.data_ptr(), b_size, d_prm.data_ptr())
    L.threadSync()
    srt = d_aos.cpu().numpy().view(np.uint32)
    B = d_B.cpu().numpy().view(np.uint32)
    assert np.array_equal(srt[:, 21], np.sort(want, kind="stable"))
    assert np.array_equal(srt[:, 21], want[srt[:, 0]])             # every struct moved whole
    assert np.array_equal(srt[:, 1:4].view(np.float32), pos[srt[:, 0]])
    z, start, cnt = np.unique(srt[:, 21], return_index=True, return_counts=True)
    ref_B = np.zeros((b_size, 2), np.uint32)
    ref_B[z, 0], ref_B[z, 1] = cnt, start
    assert np.array_equal(B, ref_B)
    # torch owns the array here, so the context behind it is dropped on its own (freeArray would free the array too)
    L.sph_compat_context.argtypes = [vp]; L.sph_compat_context.restype = vp
    L.sph_compat_release.argtypes = [vp]; L.sph_compat_release.restype = None
    assert L.sph_compat_context(d_aos.data_ptr())
    L.sph_compat_release(d_aos.data_ptr())
    assert not L.sph_compat_context(d_aos.data_ptr())


def test_seam_integers_at_config_2_in_developed_flow():
    """The same integer work at BASELINE config 2's size in a developed flow (`c2_flow`: 262,144 particles after 2600 reference
    steps, cells of 1 .. max_cell particles, chunks of 32): cudaMapZIndex / cudaSortParticles / cudaConstructBGrid /
    cudaConstructGridArray called directly on device arrays, against the oracle's Morton-mode phases (the oracle is pinned
    bit-exact to the reference's own code, tests/test_oracle_vs_ref.py): Particle::zindex per particle, the sorted z-index
    sequence, the WHOLE dev_B table (2,097,152 entries) and dev_B_prime with its size, `array_equal`."""
    import ctypes as C
    import torch
    from gpufluidsimulator_amd import capi
    from oracle import oracle
    g = load_golden("c2_flow")
    pos, vel = g["pos"], g["vel"]
    n, grid, box = pos.shape[0], int(g["grid"][0]), [float(b) for b in g["box"]]
    o = oracle.Oracle(pos, vel, g["box"], g["grid"], oracle.CELL_MORTON)
    o.map_zindex()
    want_z = o.by_index("zindex").copy()
    o.sort(); o.construct_bgrid(); o.construct_grid_array()
    want_sorted = o.particles["zindex"].copy()
    want_B = np.stack([o.B["nParticles"], o.B["start"]], axis=1).astype(np.uint32)
    want_Bp = np.stack([o.Bprime["nParticles"], o.Bprime["start"]], axis=1).astype(np.uint32)
    o.close()
    assert int(g["s1_ncells"]) == int((want_B[:, 0] > 0).sum()) and int(g["s1_max_cell"]) == int(want_B[:, 0].max())   # the fixture's own counts
    L = capi.load()
    aos = np.zeros((n, 22), np.uint32)
    aos[:, 0] = np.arange(n)
    aos[:, 1:4] = pos.view(np.uint32); aos[:, 4:7] = vel.view(np.uint32)
    aos[:, 16] = np.float32(65.0).view(np.uint32); aos[:, 19] = np.float32(1.0 / 64.0).view(np.uint32)
    prm = np.zeros(18, np.float32)
    prm[7] = 1.0 / 64.0
    prm[8:11] = [-b / 2 for b in box]; prm[11:14] = [b / 2 for b in box]; prm[14:17] = box
    prm.view(np.uint32)[17] = grid
    dev = torch.device("cuda", 0)
    d_aos, d_prm = torch.from_numpy(aos.view(np.int32)).to(dev), torch.from_numpy(prm).to(dev)
    b_size = grid ** 3
    d_B = torch.full((b_size, 2), -1, dtype=torch.int32, device=dev)
    d_Bp = torch.full((n, 2), -1, dtype=torch.int32, device=dev)
    vp = C.c_void_p
    L.cudaMapZIndex.argtypes = [vp, C.c_uint, vp]; L.cudaMapZIndex.restype = None
    L.cudaSortParticles.argtypes = [vp, C.c_uint]; L.cudaSortParticles.restype = None
    L.cudaConstructBGrid.argtypes = [vp, C.c_uint, vp, C.c_uint, vp]; L.cudaConstructBGrid.restype = None
    L.cudaConstructGridArray.argtypes = [vp, C.c_uint, vp, C.c_uint, C.POINTER(vp), C.POINTER(C.c_uint), vp]
    L.cudaConstructGridArray.restype = None
    L.sph_compat_release.argtypes = [vp]; L.sph_compat_release.restype = None
    L.threadSync.restype = None
    try:
        L.cudaMapZIndex(d_aos.data_ptr(), n, d_prm.data_ptr())
        L.threadSync()
        assert np.array_equal(d_aos.cpu().numpy().view(np.uint32)[:, 21], want_z)
        L.cudaSortParticles(d_aos.data_ptr(), n)
        L.cudaConstructBGrid(d_aos.data_ptr(), n, d_B.data_ptr(), b_size, d_prm.data_ptr())
        bp, bp_size = vp(d_Bp.data_ptr()), C.c_uint(0)
        L.cudaConstructGridArray(d_aos.data_ptr(), n, d_B.data_ptr(), b_size, C.byref(bp), C.byref(bp_size), d_prm.data_ptr())
        L.threadSync()
        srt = d_aos.cpu().numpy().view(np.uint32)
        assert np.array_equal(srt[:, 21], want_sorted)
        assert np.array_equal(np.sort(srt[:, 0]), np.arange(n)) and np.array_equal(srt[:, 21], want_z[srt[:, 0]])
        assert np.array_equal(d_B.cpu().numpy().view(np.uint32), want_B)
        assert bp_size.value == want_Bp.shape[0]
        assert np.array_equal(d_Bp.cpu().numpy().view(np.uint32)[:bp_size.value], want_Bp)
    finally:
        L.sph_compat_release(d_aos.data_ptr())
