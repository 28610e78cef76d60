#!/usr/bin/env python3
"""BASELINE config 3 at its full size in a FREE run against the reference's own code (oracle/_ref/sph_ref, OpenMP mode):
the device brings the dam into its flowing state (2600 steps), hands that state to the reference, and both run `steps` more
steps on their own; every `every` steps all 16,777,216 particles are compared.
    python profiles/scripts/c3_free_run_vs_reference.py [steps] [every]       (GPU box, repo root)
Prints max |dpos| / box, max |dvel| / |v|max, max |drho / rho| and the particles beyond 1e-5 / 1e-4 of |v|max (a free run
parts at discrete events -- a pair that counts as colliding in one run and not in the other: DESIGN.md section 4)."""
import os
import sys
import time

import numpy as np

ROOT = os.getcwd()
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from gpufluidsimulator_amd import capi, ic  # noqa: E402
from oracle import refio  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
every = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = ic.CONFIGS["C3"]
n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
dt = float(ic.DEFAULT_DT)
threads = max(1, min(32, os.cpu_count() or 1))
dumps = tuple(range(every, steps + 1, every))
with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.reset_lattice(cfg["lattice"], jitter=True)
    c.step(dt, 2600)
    s0 = c.download(want=("pos", "vel"))
    print(f"config 3, {n} particles, flowing state after 2600 steps (movers so far {c.sort_stats()['movers_total']}); "
          f"free run of {steps} steps against the reference's own code ({threads} threads)", flush=True)
    t0 = time.time()
    recs, st = refio.run_ref(s0["pos"], s0["vel"], cfg["box"], cfg["grid"][0], dt, steps, dump_steps=dumps, threads=threads)
    print(f"reference: {time.time() - t0:.1f} s for {steps} steps ({st.get('particle_steps_per_s', 0):.3g} particle-steps/s)", flush=True)
    done = 0
    for k in dumps:
        c.step(dt, k - done); done = k
        a, b = c.download(), recs[("state", k)]
        vmax = float(np.abs(b[:, 3:6]).max())
        ev = np.abs(a["vel"] - b[:, 3:6]).max(axis=1) / vmax
        print("  step %3d  dpos/box %.2e  dvel/|v|max %.2e  drho/rho %.2e  particles beyond 1e-5: %d, beyond 1e-4: %d (of %d)" %
              (k, float(np.abs(a["pos"] - b[:, 0:3]).max() / cfg["box"][0]), float(ev.max()),
               float(np.abs(a["density"] / b[:, 6] - 1).max()), int((ev > 1e-5).sum()), int((ev > 1e-4).sum()), n), flush=True)
