import sys, os, ctypes
import torch  # noqa
sys.path.insert(0, os.getcwd())
from gpufluidsimulator_amd import capi, ic
cfg = ic.CONFIGS["C3"]
n = 256 ** 3
runup = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.reset_lattice(cfg["lattice"], jitter=True)
    lib = capi.load()
    out = (ctypes.c_ulonglong * 12)()
    for label, steps in (("rest", 3), ("flow", runup)):
        c.step(5e-7, steps); c.sync()
        lib.sph_debug_pair_stats(out, 1)
        c.step(5e-7, 1); c.sync()
        lib.sph_debug_pair_stats(out, 0)
        w = out[0]
        print(label, "waves", w, "pieces/wave %.2f" % (out[1] / w), "walk T/wave %.1f" % (out[2] / w), "chunks/wave %.2f" % (out[3] / w),
              "coll rounds/wave %.2f" % (out[4] / w), "waves with a hull > 256 / 512 / 2048 slots: %d / %d / %d" % (out[5], out[6], out[7]),
              "| collision candidates (superset) per particle %.2f, of the busiest lane of a wave %.2f (= rounds per wave of a per-particle queue)" % (out[9] / (64.0 * w), out[8] / w))
