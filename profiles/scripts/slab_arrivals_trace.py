"""Two slabs of one GPU, every particle moving up: arrivals from about step 60 on.  Run under rocprofv3 --kernel-trace."""
import json, os, sys, threading
import numpy as np
import torch  # noqa
sys.path.insert(0, os.getcwd())
from gpufluidsimulator_amd import ic, slab
world, steps = 2, int(sys.argv[1])
cfg = ic.weak_scaling_config(world, per_gpu=(160, 160, 160))
pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=True, jitter_dims=cfg["jitter_dims"])
vel[:, 2] = 600.0
hub = slab.LocalComm.Hub(world)
out = [None] * world
def rank_main(r):
    sim = slab.NativeSlabSimulation(slab.LocalComm(hub, r), cfg["box"], cfg["grid"], device_index=0, transport="host",
                                    particles=(pos, vel))
    sim.run(5e-7, steps); sim.sync()
    out[r] = {k: sim.stats.get(k, 0) for k in ("migrants", "resorts", "in_place_merges", "steps")}
    sim.close()
ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
print(json.dumps({"lib": os.path.basename(os.environ.get("SPH_HIP_LIB", "default")), "per_rank": out}))
