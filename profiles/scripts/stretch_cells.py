#!/usr/bin/env python3
"""Follow-up of stretch_experiments.py: the slow state of the thin lattice is slow because of its POSITIONS relative to
the cell grid (a shift by 0.01 makes it fast, noise does not).  For the fast state, the slow state and the shifted
slow state: cell-occupancy histogram, candidates per particle, the wave-uniform walk, and -- from a -DSPH_PAIR_STATS
build of the library (SPH_HIP_LIB=scratch/v/libsph_stats.so) -- the pieces, walk length and chunks per wave that the
force kernel actually ran, plus the hull lengths per wave computed on the host from the keys and the cell table.
    SPH_HIP_LIB=scratch/v/libsph_stats.so python profiles/scripts/stretch_cells.py [slow_step]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from gpufluidsimulator_amd import capi, ic  # noqa: E402

DT = float(ic.DEFAULT_DT)
cfg = ic.CONFIGS["C3"]
lattice = (256, 256, 32)
n = lattice[0] * lattice[1] * lattice[2]
slow_step = int(sys.argv[1]) if len(sys.argv) > 1 else 1780
gx, gy, gz = cfg["grid"]


def analyse(label, pos, vel):
    lib = capi.load()
    with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
        c.upload(np.ascontiguousarray(pos), np.ascontiguousarray(vel))
        c.hash(); c.sort(); c.build_cells(); c.density(); c.sync()
        t0 = time.perf_counter()
        for _ in range(50):
            c.density()
        c.sync()
        ms_d = (time.perf_counter() - t0) / 50 * 1e3
        stats = None
        if hasattr(lib, "sph_debug_pair_stats"):
            out = (ctypes.c_ulonglong * 12)()
            lib.sph_debug_pair_stats(out, 1)
            c.force(); c.sync()
            lib.sph_debug_pair_stats(out, 0)
            w = max(out[0], 1)
            stats = f"waves {out[0]} pieces/wave {out[1] / w:.2f} walk T/wave {out[2] / w:.1f} chunks/wave {out[3] / w:.2f}"
        t0 = time.perf_counter()
        for _ in range(20):
            c.force()
        c.sync()
        ms_f = (time.perf_counter() - t0) / 20 * 1e3
        k, s, cnt = c.cells(max_cells=c.n)
        keys = c.keys().astype(np.int64)
    hist = np.bincount(cnt, minlength=20)
    # hull length per wave and row, as the kernels build it: lo of the first lane with a range, hi of the last
    start = np.zeros(gx * gy * gz + 2, dtype=np.int64); end = np.zeros_like(start)
    start[k + 1] = s; end[k + 1] = s + cnt                           # +1: guard entry in front
    nw = keys.size // 64
    kw = keys[: nw * 64].reshape(nw, 64)
    cx, cy, cz = kw % gx, (kw // gx) % gy, kw // (gx * gy)
    hull_tot = np.zeros(nw, dtype=np.int64)
    walk = np.zeros(nw, dtype=np.int64)
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            ok = (cy + dy >= 0) & (cy + dy < gy) & (cz + dz >= 0) & (cz + dz < gz)
            kk = np.where(ok, kw + (dz * gy + dy) * gx, kw)
            lo = np.full(kw.shape, np.iinfo(np.int64).max); hi = np.zeros(kw.shape, dtype=np.int64); ln = np.zeros(kw.shape, dtype=np.int64)
            for dx in (-1, 0, 1):
                okx = ok & (cx + dx >= 0) & (cx + dx < gx)
                st_, en_ = start[kk + dx + 1], end[kk + dx + 1]
                has = okx & (en_ > st_)
                lo = np.where(has, np.minimum(lo, st_), lo)
                hi = np.where(has, np.maximum(hi, en_), hi)
                ln += np.where(has, en_ - st_, 0)
            any_ = hi > 0
            A = np.where(any_, lo, np.iinfo(np.int64).max).min(axis=1)
            B = hi.max(axis=1)
            hull_tot += np.where(B > 0, B - A, 0)
            walk += ln.max(axis=1)
    pieces = None
    print(f"{label}: density {ms_d:.3f} ms, force {ms_f:.3f} ms | cells {k.size}, per cell mean {cnt.mean():.2f} max {cnt.max()} hist 1..16 {hist[1:17].tolist()}"
          f" | walk/wave mean {walk.mean():.1f} max {walk.max()} | hull entries per wave (9 rows) mean {hull_tot.mean():.0f} "
          f"median {np.median(hull_tot):.0f} 99% {np.quantile(hull_tot, .99):.0f} max {hull_tot.max()} | waves with hull > 2000: {(hull_tot > 2000).sum()} of {nw},"
          f" their share of all hull entries {hull_tot[hull_tot > 2000].sum() / hull_tot.sum():.2f} | {stats}", flush=True)
    # where are the long-hull waves?
    big = np.nonzero(hull_tot > 2000)[0]
    if big.size:
        b = big[:: max(1, big.size // 6)][:6]
        for w in b:
            print(f"     wave {w}: hull {hull_tot[w]}, keys span cells x {cx[w].min()}..{cx[w].max()} y {cy[w].min()}..{cy[w].max()} z {cz[w].min()}..{cz[w].max()}", flush=True)


with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.reset_lattice(lattice, jitter=True)
    c.step(DT, 805); c.sync()
    fast = c.download(want=("pos", "vel"))
    c.step(DT, slow_step - 805); c.sync()
    slow = c.download(want=("pos", "vel"))
analyse("S_fast (805)", fast["pos"], fast["vel"])
analyse(f"S_slow ({slow_step})", slow["pos"], slow["vel"])
analyse("S_slow + 0.01", slow["pos"] + np.float32(0.01), slow["vel"])
for ax, nm in enumerate("xyz"):
    p = slow["pos"].copy(); p[:, ax] += np.float32(0.01)
    analyse(f"S_slow + 0.01 in {nm} only", p, slow["vel"])
