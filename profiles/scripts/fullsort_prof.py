import sys, os
import torch  # noqa
sys.path.insert(0, os.getcwd())
from gpufluidsimulator_amd import capi, ic
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = ic.CONFIGS[wl]
n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.set_sort_mode(0)
    c.reset_lattice(cfg["lattice"], jitter=True)
    c.step(5e-7, 30); c.sync()
