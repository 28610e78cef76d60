"""Where does a free GPU run leave the 1e-5 velocity bar, and is a collision predicate involved?
GPU (phase API, so that collision counts can be read every step) and the CPU oracle (bit-exact with the reference
binary) both run FREELY from the same initial state; per step: velocity error relative to |v|max, number of particles
whose collision COUNT differs, and -- at the first step where a particle exceeds 1e-5 -- that particle's counts and
delta_v on both sides.  Run once per library variant (SPH_HIP_LIB)."""
import json, os, sys
import numpy as np
import torch  # noqa
sys.path.insert(0, os.getcwd())
from gpufluidsimulator_amd import capi
from oracle import oracle
from tests.conftest import load_golden

name = sys.argv[1] if len(sys.argv) > 1 else "c1_jitter"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g = load_golden(name)
dt = float(g["dt"])
o = oracle.Oracle(g["pos"], g["vel"], g["box"], g["grid"], oracle.CELL_LINEAR)
out = {"lib": os.path.basename(os.environ.get("SPH_HIP_LIB", "default")), "case": name, "first_over_1e-5": None,
       "count_mismatch_steps": [], "per_step": []}
with capi.Context(g["pos"].shape[0], box=g["box"], grid=g["grid"]) as c:
    c.upload(g["pos"], g["vel"])
    for s in range(1, steps + 1):
        c.hash(); c.sort(); c.build_cells(); c.density(); c.force(); c.collide()
        f = c.download_forces(force=False)
        o.map_zindex(); o.sort(); o.apply_order(c.order()); o.construct_bgrid()
        o.compute_densities(); o.compute_forces(); o.particle_collisions()
        oc, odv = o.by_index("collision_count"), o.by_index("delta_velocity")
        mism = np.nonzero(f["count"] != oc)[0]
        c.integrate(dt); o.integrate(dt)
        st, so = c.download(want=("vel",)), o.state()
        ev = np.abs(st["vel"] - so["vel"]).max(axis=1) / np.abs(so["vel"]).max()
        rec = {"step": s, "vel_err_max": float(ev.max()), "over_1e-5": int((ev > 1e-5).sum()), "count_mismatch": int(mism.size)}
        if mism.size:
            out["count_mismatch_steps"].append(s)
        if out["first_over_1e-5"] is None and (ev > 1e-5).any():
            w = int(np.argmax(ev))
            out["first_over_1e-5"] = {"step": s, "particle": w, "err": float(ev[w]), "gpu_count": int(f["count"][w]),
                                      "oracle_count": int(oc[w]), "gpu_dv": f["dv"][w].tolist(), "oracle_dv": odv[w].tolist(),
                                      "count_mismatch_particles_this_step": mism[:8].tolist(),
                                      "count_mismatch_in_earlier_steps": list(out["count_mismatch_steps"])}
        if s in (1, 2, 5, 10, 20, 50, 100) or rec["over_1e-5"] or rec["count_mismatch"]:
            out["per_step"].append(rec)
        if s in (1, 10, 100) and f"state_{s}" in g.files:
            ref = g[f"state_{s}"]
            eg = np.abs(st["vel"] - ref[:, 3:6]).max(axis=1) / np.abs(ref[:, 3:6]).max()
            out[f"vs_reference_golden_step_{s}"] = {"vel_err_max": float(eg.max()), "over_1e-5": int((eg > 1e-5).sum()),
                                                     "median": float(np.median(eg))}
print(json.dumps(out))
