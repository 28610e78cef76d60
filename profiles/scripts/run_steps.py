#!/usr/bin/env python3
"""C3 from the lattice, N steps, nothing else (a target for rocprofv3 --pmc / --kernel-trace over a stretch of the run-up).
    python profiles/scripts/run_steps.py <steps>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from gpufluidsimulator_amd import capi, ic  # noqa: E402

cfg = ic.CONFIGS["C3"]
n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.reset_lattice(cfg["lattice"], jitter=True)
    c.step(float(ic.DEFAULT_DT), int(sys.argv[1]))
    c.sync()
