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
    g = load_golden("random_clump")
    recs, _ = refio.run_ref(g["pos"], g["vel"], g["box"], int(g["grid"][0]), float(g["dt"]), 2,
                            dump_steps=(1, 2), binary=refio.DROPIN_BIN)
    for s in (1, 2):
        st, ref = recs[("state", s)], g[f"s{s}_state"]
        assert np.abs(st[:, 0:3] - ref[:, 0:3]).max() <= 4e-6 * 2.0
        assert np.abs(st[:, 6] / ref[:, 6] - 1).max() <= 1e-5
        bad = np.abs(st[:, 3:6] - ref[:, 3:6]).max(axis=1) > 4e-5 * np.abs(ref[:, 3:6]).max()
        assert bad.mean() <= 2e-3
