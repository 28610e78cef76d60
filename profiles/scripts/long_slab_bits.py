#!/usr/bin/env python3
"""A LONG slab run against one context, bit for bit: BASELINE config 2's dam in `world` slabs (threads of one process over the
device-to-device transport), `steps` steps with re-balancing checks every 500.
    python profiles/scripts/long_slab_bits.py [world] [steps] [protocol: 3 | 1]      (GPU box, repo root)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch  # noqa: F401,E402
from gpufluidsimulator_amd import capi, ic  # noqa: E402
import test_gpu_slabs as T  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20100
protocol = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = ic.CONFIGS["C2"]
t0 = time.time()
res = T._run_slabs(world, cfg["box"], cfg["grid"], steps, lattice=cfg["lattice"], rebalance_every=500, protocol=protocol)
t1 = time.time()
pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=True)
with capi.Context(pos.shape[0], box=cfg["box"], grid=cfg["grid"]) as c:
    c.upload(pos, vel)
    c.step(T.DT, steps)
    ref = c.download()
    movers = c.sort_stats()["movers_total"]
st = res[0][0]
same = {k: bool(np.array_equal(st[k].view(np.uint32), ref[k].view(np.uint32))) for k in ("pos", "vel", "density", "pressure")}
stats = {k: sum(r[1].get(k, 0) for r in res) for k in ("migrants", "resorts", "in_place_merges", "far_steps", "rest_messages", "rebalances",
                                                       "one_message_steps", "one_message_rests")}
print(f"C2 dam, {world} slabs, protocol {protocol}, {steps} steps ({t1 - t0:.0f} s): bit-identical to the one-context run: {same}; cuts at the end {res[0][2]}; "
      f"{stats}; movers of the one-context run {movers} ({movers / pos.shape[0]:.1f} per particle); |v|max {np.abs(ref['vel']).max():.1f}", flush=True)
sys.exit(0 if all(same.values()) else 1)
