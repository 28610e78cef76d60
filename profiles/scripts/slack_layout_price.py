#!/usr/bin/env python3
"""What a SLACK layout of the sorted arrays could save, measured on the flowing C3 dam.

`k_mm_move` streams all 16.7 M particles (1.2 GB at the achievable HBM rate, ~205 us) to place ~10^4 movers.  The
alternative data structure: chunks of G slots that each keep a few FREE slots, so that a particle that changes cell only
rewrites the chunk it leaves and the chunk it joins.  Its price list needs three numbers that only the running flow
can give, and this script measures them on the device (the bench's state: C3 after `runup` steps, then K steps):
  dirty(G)   -- share of chunks that gain, lose or re-order a particle in a step: the bytes a slack layout still moves;
  slack(G,K) -- free slots per chunk needed to survive K steps without re-packing: the largest net inflow of any chunk
                (max, and the 99.9th percentile: a few overflowing chunks could spill into a neighbour);
  and the cost of holes: every free slot is a slot the pair kernels stage, walk (zero weight, full instruction cost)
  and give an idle lane -- slack / G of (k_density + k_force), whose times the caller passes in.
Chunks are runs of whole cells holding ~G slots at the re-pack (fixed key boundaries afterwards).
    python profiles/scripts/slack_layout_price.py [runup=6000] [K=200] [dens_us=890] [force_us=2440] [move_us=205]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: F401,E402
from gpufluidsimulator_amd import capi, ic  # noqa: E402

runup = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dens_us, force_us, move_us = (float(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((3, 890.0), (4, 2440.0), (5, 205.0)))
cfg = ic.CONFIGS["C3"]
n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
DT = float(ic.DEFAULT_DT)
GS = (64, 256, 1024, 4096)


def keys_by_index(c):
    c.hash(); c.sort()                      # (the first phases of the next step: idempotent for the step that follows)
    k, o = c.keys(), c.order()
    out = np.empty(n, np.uint32)
    out[o] = k
    return out, k


with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.reset_lattice(cfg["lattice"], jitter=True)
    t0 = time.time()
    c.step(DT, runup); c.sync()
    print(f"C3, {n} particles, {runup} run-up steps in {time.time() - t0:.1f} s; then {K} steps, keys by creation index after every sort", flush=True)
    key0, sorted_keys = keys_by_index(c)
    bounds = {G: sorted_keys[::G].copy() for G in GS}             # first key of every chunk at the re-pack
    chunk_of = {G: (lambda kk, B=bounds[G]: np.searchsorted(B, kk, side="right") - 1) for G in GS}
    nchunks = {G: bounds[G].shape[0] for G in GS}
    delta = {G: np.zeros(nchunks[G], np.int64) for G in GS}        # occupancy now - occupancy at the re-pack
    peak = {G: np.zeros(nchunks[G], np.int64) for G in GS}         # the largest it has been so far
    dirty = {G: [] for G in GS}
    movers = []
    marks = {}
    prev = key0
    for step in range(1, K + 1):
        c.step(DT, 1)
        cur, _ = keys_by_index(c)
        mv = np.nonzero(cur != prev)[0]
        movers.append(mv.size)
        for G in GS:
            a, b = chunk_of[G](prev[mv]), chunk_of[G](cur[mv])
            dirty[G].append(np.unique(np.concatenate([a, b])).size / nchunks[G])
            delta[G] += np.bincount(b, minlength=nchunks[G]) - np.bincount(a, minlength=nchunks[G])
            np.maximum(peak[G], delta[G], out=peak[G])
            if step in (25, 50, 100, 200, K):
                marks[(G, step)] = (int(peak[G].max()), float(np.percentile(peak[G], 99.9)), float((peak[G] > 0).mean()))
        prev = cur
    mv = np.array(movers)
    print(f"movers per step: mean {mv.mean():.0f} ({mv.mean() / n:.2e} N), median {np.median(mv):.0f}, max {mv.max()}", flush=True)
    print("G = slots per chunk | dirty chunks per step: mean, median, max | slack needed after 25 / 50 / 100 / 200 steps: max (99.9th pct) | "
          "price of that slack at 50 steps (us per step) against what the chunk-wise move saves (us per step)")
    for G in GS:
        d = np.array(dirty[G])
        need = " ; ".join(f"{marks[(G, s)][0]} ({marks[(G, s)][1]:.0f})" for s in (25, 50, 100, 200) if (G, s) in marks)
        s50 = marks.get((G, 50), marks[(G, K)])
        # a chunk must hold its 99.9th-percentile inflow (the rest spill): slack fraction of every staged / walked / idle slot
        frac = s50[1] / G
        cost = frac * (dens_us + force_us)
        save = (1.0 - d.mean()) * move_us
        print(f"G = {G:5d} | {d.mean():.3f}, {np.median(d):.3f}, {d.max():.3f} | {need} | holes {frac * 100:.1f} % of the slots -> "
              f"+{cost:.0f} us in the pair kernels ; move saves at most {save:.0f} us", flush=True)
