"""Would an HBM-bound move overlap with the VALU-bound density pass?  C3 flowing state: k_density on the library's stream, a 1.2 GB
device copy (the traffic of k_mm_move) on a torch stream, alone and together."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gpufluidsimulator_amd import capi, ic
cfg = ic.CONFIGS["C3"]; n = int(np.prod(cfg["lattice"])); dt = float(ic.DEFAULT_DT)
a = torch.empty(600 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
s2 = torch.cuda.Stream()
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.reset_lattice(cfg["lattice"], jitter=True); c.step(dt, 3000); c.sync()
    c.hash(); c.sort(); c.build_cells()
    def dens(): c.density(); 
    def copy():
        with torch.cuda.stream(s2): b.copy_(a, non_blocking=True)
    def both(): copy(); dens()
    def both2(): dens(); copy()
    def dens_sync(): c.density(); c.sync()
    print("density alone  %.3f ms" % timeit(lambda: (dens(), c.sync())))
    print("copy alone     %.3f ms (600 MB read + 600 MB write)" % timeit(lambda: (copy(), s2.synchronize())))
    print("copy then dens %.3f ms (two streams, both synchronised)" % timeit(lambda: (both(), c.sync(), s2.synchronize())))
    print("dens then copy %.3f ms" % timeit(lambda: (both2(), c.sync(), s2.synchronize())))
    c.force()
    print("force-only alone %.3f ms" % timeit(lambda: (c.force(), c.sync())))
    print("copy + force-only %.3f ms" % timeit(lambda: (copy(), c.force(), c.sync(), s2.synchronize())))
