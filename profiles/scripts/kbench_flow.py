#!/usr/bin/env python3
"""Per-kernel A/B of library variants in the FLOWING state of the bench (C3 after 6000 steps): the state is prepared once
(default library) and saved as a snapshot under /tmp; every variant runs in its own process, loads it, steps 10 + 40 times
with the device timers on and prints its per-phase means.
    python profiles/scripts/kbench_flow.py scratch/v/libsph_A.so scratch/v/libsph_B.so ...   (GPU box, repo root)"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.getcwd())
SNAP = os.environ.get("KB_SNAP", "/tmp/c3_flow.snap")


def one():
    import numpy as np
    import torch  # noqa: F401
    from gpufluidsimulator_amd import capi, ic
    cfg = ic.CONFIGS["C3"]
    n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
    dt = float(ic.DEFAULT_DT)
    with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
        c.set_precision(bool(os.environ.get("KB_MIXED")))          # KB_MIXED=1: BASELINE config 5's arithmetic (k_density_h)
        if sys.argv[2] == "prepare":
            c.reset_lattice(cfg["lattice"], jitter=True)
            c.step(dt, 6000); c.sync()
            c.save(SNAP)
            return
        c.load_snapshot(SNAP)
        c.step(dt, 10); c.sync()
        c.timing(True); c.timing_reset()
        t0 = time.perf_counter()
        c.step(dt, 40); c.sync()
        wall = (time.perf_counter() - t0) / 40 * 1e3
        ph, k = c.timing_get()
        st = c.download(want=("density", "vel"))
    print(json.dumps({"lib": os.path.basename(os.environ.get("SPH_HIP_LIB", "default")), "ms_per_step": round(wall, 4),
                      **{p: round(v / k, 4) for p, v in ph.items() if v},
                      "rho_sum": float(st["density"].astype(np.float64).sum()), "v_abs": float(np.abs(st["vel"].astype(np.float64)).sum())}))


if len(sys.argv) > 1 and sys.argv[1] == "--one":
    one()
else:
    if not os.path.exists(SNAP):
        subprocess.run([sys.executable, __file__, "--one", "prepare"], check=True, timeout=600)
    for lib in sys.argv[1:]:
        env = dict(os.environ, SPH_HIP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__, "--one", "run"], env=env, capture_output=True, text=True, timeout=600)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + lib + " " + r.stderr[-600:]), flush=True)
