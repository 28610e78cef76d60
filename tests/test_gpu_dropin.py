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


def _morton_decode(z):
    """zIndex2coord of the reference (particleSystem.cu:105-124): every third bit."""
    z = np.asarray(z, dtype=np.uint64)
    out = []
    for a in range(3):
        v = np.zeros_like(z)
        for b in range(10):
            v |= ((z >> np.uint64(3 * b + a)) & np.uint64(1)) << np.uint64(b)
        out.append(v.astype(np.int64))
    return out


@pytest.mark.parametrize("name", ["c1_jitter", "random_clump"])
def test_seam_fills_B_and_Bprime_like_the_reference(name):
    """cudaConstructBGrid / cudaConstructGridArray leave {nParticles, start} per occupied cell in the caller's B and
    one {nParticles <= 32, start} entry per 32-particle chunk in B', with the size handed back to the host
    (particleSystem.cu:503-528).  Cells carry this library's row-major number (as does Particle::zindex), so B is
    compared cell by cell through the cell coordinates; B' must chunk the sorted array exactly like the
    reference's (same number of entries, same chunk sizes per cell)."""
    g = load_golden(name)
    grid = int(g["grid"][0])
    recs, _ = refio.run_ref(g["pos"], g["vel"], g["box"], grid, float(g["dt"]), 1, phases=True, binary=refio.DROPIN_BIN)
    cells, bprime = recs[("bcells", 1)], recs[("bprime", 1)]
    sorted_z, order = recs[("sorted_z", 1)], recs[("order", 1)]
    ref_cells, ref_bprime = g["s1_bcells"], g["s1_bprime"]
    n = order.shape[0]
    # the particle array is sorted by the key it carries, and B describes exactly its runs
    assert np.all(np.diff(sorted_z.astype(np.int64)) >= 0)
    key, cnt, start = cells[:, 0].astype(np.int64), cells[:, 1].astype(np.int64), cells[:, 2].astype(np.int64)
    assert cnt.sum() == n
    for k, c, s in zip(key[:2000], cnt[:2000], start[:2000]):
        assert np.all(sorted_z[s:s + c] == k) and (s == 0 or sorted_z[s - 1] != k) and (s + c == n or sorted_z[s + c] != k)
    # same occupied cells with the same counts as the reference's B (Morton-numbered there)
    x, y, z = key % grid, (key // grid) % grid, key // (grid * grid)
    rx, ry, rz = _morton_decode(ref_cells[:, 0])
    mine = dict(zip(zip(x.tolist(), y.tolist(), z.tolist()), cnt.tolist()))
    theirs = dict(zip(zip(rx.tolist(), ry.tolist(), rz.tolist()), ref_cells[:, 1].astype(np.int64).tolist()))
    assert mine == theirs
    # B': one entry per chunk of <= 32, tiling the sorted array in order; same multiset of chunk sizes per cell
    assert bprime.shape[0] == ref_bprime.shape[0]
    bs, bn = bprime[:, 0].astype(np.int64), bprime[:, 1].astype(np.int64)
    assert bs[0] == 0 and np.all(bs[1:] == bs[:-1] + bn[:-1]) and bs[-1] + bn[-1] == n and bn.max() <= 32 and bn.min() >= 1
    assert sorted(bn.tolist()) == sorted(ref_bprime[:, 1].astype(np.int64).tolist())
    for s, c in zip(bs[:2000], bn[:2000]):
        assert np.all(sorted_z[s:s + c] == sorted_z[s])
