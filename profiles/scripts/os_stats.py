import sys, os, ctypes
import torch  # noqa
sys.path.insert(0, os.getcwd())
from gpufluidsimulator_amd import capi, ic
cfg = ic.CONFIGS["C3"]
n = 256 ** 3
with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.set_sort_mode(0)
    c.reset_lattice(cfg["lattice"], jitter=True)
    c.step(5e-7, 3); c.sync()
    lib = capi.load()
    out = (ctypes.c_ulonglong * 10)()
    lib.sph_debug_os_stats(out, 1)
    c.step(5e-7, 4); c.sync()
    lib.sph_debug_os_stats(out, 1)
    tot = sum(out)
    names = ["between", "ticket", "load+count", "publish+scan", "rank", "lookback", "sync", "writeout", "endsync", "-"]
    print({k: round(100.0 * v / tot, 1) for k, v in zip(names, out)}, "ticks per tile", tot / (4 * 3 * 4096))
