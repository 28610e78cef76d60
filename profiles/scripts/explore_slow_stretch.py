#!/usr/bin/env python3
"""What do the cells look like in the stretch of the C3 run-up where k_force / k_density run 1.7-1.9x slower than before
and after (steps ~1700-1800: profiles/r03_c3_outlier_launches.txt)?  Steps the device to a few step numbers, downloads the
cell table and reports the cell-occupancy histogram, the candidates per particle (27-cell sums) and the quantity the pair
kernels' time follows: per wave of 64 consecutive sorted particles and per (dz, dy) row, the LONGEST lane's 3-cell sum
(the wave-uniform trip count), summed over the 9 rows.

    python profiles/scripts/explore_slow_stretch.py [step ...]      (run on the GPU box from the repo root)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from gpufluidsimulator_amd import capi, ic  # noqa: E402


def report(ctx, cfg, step, ms):
    gx, gy, gz = cfg["grid"]
    k, s, cnt = ctx.cells(max_cells=ctx.n)
    occ = np.zeros(gx * gy * gz, dtype=np.int32)
    occ[k] = cnt
    occ = occ.reshape(gz, gy, gx)
    hist = np.bincount(cnt, minlength=20)
    # 3-cell sums along x (a lane's range in one row), then the 27-cell sum per cell
    pad = np.pad(occ, 1)
    row3 = pad[:, :, :-2] + pad[:, :, 1:-1] + pad[:, :, 2:]                     # (gz+2, gy+2, gx): sum over dx
    cand = np.zeros_like(occ, dtype=np.int64)
    for dz in range(3):
        for dy in range(3):
            cand += row3[dz:dz + gz, dy:dy + gy, :]
    per_particle = float((cand * occ).sum()) / float(occ.sum())
    # wave-uniform walk: 64 consecutive particles; per row the max over the wave's lanes of the 3-cell sum
    keys = ctx.keys()
    sub = slice(0, min(keys.size, 1 << 22))                                      # the first 4 M sorted particles
    kk = keys[sub].astype(np.int64)
    cz, cy, cx = kk // (gx * gy), (kk // gx) % gy, kk % gx
    walk = np.zeros(kk.size // 64, dtype=np.int64)
    for dz in range(3):
        for dy in range(3):
            v = row3[cz + dz, cy + dy, cx]
            walk += v[: walk.size * 64].reshape(-1, 64).max(axis=1)
    print(f"step {step:5d}: {ms:6.3f} ms/step (fused step, last 10) | occupied cells {k.size}, particles per occupied cell mean {cnt.mean():.2f} "
          f"max {cnt.max()} | histogram of counts 1..16: {hist[1:17].tolist()} | candidates per particle {per_particle:.1f} | "
          f"wave-uniform walk (sum over 9 rows of the longest lane) mean {walk.mean():.1f} max {walk.max()}", flush=True)


def main():
    steps = [int(v) for v in sys.argv[1:]] or [900, 1080, 1400, 1650, 1730, 1800, 2000, 3700]
    cfg = ic.CONFIGS["C3"]
    n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
    dt = float(ic.DEFAULT_DT)
    with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
        c.reset_lattice(cfg["lattice"], jitter=True)
        done = 0
        for s in steps:
            c.step(dt, s - done - 10); c.sync()
            t0 = time.perf_counter()
            c.step(dt, 10); c.sync()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            done = s
            c.hash(); c.sort(); c.build_cells()       # the table of the CURRENT positions (phase API)
            report(c, cfg, s, ms)


if __name__ == "__main__":
    main()
