#!/usr/bin/env python3
"""How many tiles of the sorted arrays keep exactly their occupants from one step to the next?  ("do
not move the non-movers" -- k_mm_move rewrites all 16.7 M particles for 0.03-0.2 % movers.)  A tile whose slots hold the
same particles in the same order after the sort needs no move at all.  Flowing C3 after `runup` steps, `count` steps:
per step the movers and the share of static tiles at 1024 / 4096 / 16384 / 65536 slots per tile.
    python profiles/scripts/static_tiles.py [runup] [count]      (GPU box, repo root)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: F401,E402
from gpufluidsimulator_amd import capi, ic  # noqa: E402

runup = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = ic.CONFIGS["C3"]
n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
dt = float(ic.DEFAULT_DT)
with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.reset_lattice(cfg["lattice"], jitter=True)
    c.step(dt, runup); c.sync()
    _, _, prev = c.download_owned()
    q0 = c.sort_stats()["movers_total"]
    for s in range(count):
        c.step(dt, 1); c.sync()
        _, _, idx = c.download_owned()          # slot order AFTER this step's sort... (the state is sorted at the start of a step)
        q1 = c.sort_stats()["movers_total"]
        same = prev == idx
        out = []
        for t in (1024, 4096, 16384, 65536):
            k = n // t
            out.append(f"{t}: {same[: k * t].reshape(k, t).all(axis=1).mean() * 100:5.1f} %")
        print(f"step {runup + s + 1}: movers {q1 - q0:7d}, slots that keep their particle {same.mean() * 100:5.1f} %, static tiles  " + "  ".join(out), flush=True)
        prev, q0 = idx, q1
