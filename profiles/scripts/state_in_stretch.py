#!/usr/bin/env python3
"""Physical state of C3 around the slow stretch of its run-up: density and pressure statistics at a few steps."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from gpufluidsimulator_amd import capi, ic  # noqa: E402

cfg = ic.CONFIGS["C3"]
n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
dt = float(ic.DEFAULT_DT)
with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.reset_lattice(cfg["lattice"], jitter=True)
    done = 0
    for s in (1000, 1500, 1650, 1700, 1720, 1740, 1760, 1800, 1900, 2400):
        c.step(dt, s - done - 5); c.sync()
        t0 = time.perf_counter(); c.step(dt, 5); c.sync(); ms = (time.perf_counter() - t0) / 5 * 1e3
        done = s
        st = c.download(want=("density", "pressure", "vel"))
        rho, p, v = st["density"], st["pressure"], st["vel"]
        q = np.quantile(rho, [0.001, 0.01, 0.5, 0.99, 0.999])
        print(f"step {s:5d}: {ms:6.3f} ms/step | density quantiles 0.1/1/50/99/99.9 %: {q.round(1).tolist()} | particles with p > 0: "
              f"{(p > 0).mean() * 100:6.2f} % | pressure median {np.median(p):.1f} max {p.max():.1f} | |v|max {np.abs(v).max():.1f} "
              f"vy median {np.median(v[:, 1]):.1f}", flush=True)
