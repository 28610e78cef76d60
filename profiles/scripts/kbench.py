"""Per-kernel timing of lib variants on C3 (or C2 with --small): one process per variant."""
import sys, os, json, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    from gpufluidsimulator_amd import capi, ic
    cfg = ic.CONFIGS[os.environ.get("KB_CFG", "C3")]
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=True)
    with capi.Context(pos.shape[0], box=cfg["box"], grid=cfg["grid"]) as c:
        c.upload(pos, vel)
        c.step(5e-7, 3); c.sync()
        c.timing(True); c.timing_reset()
        c.step(5e-7, 8); c.sync()
        ph, n = c.timing_get()
        st = c.download(want=("density", "vel"))
    print(json.dumps({"lib": os.path.basename(os.environ.get("SPH_HIP_LIB", "default")), **{k: round(v / n, 4) for k, v in ph.items() if v},
                      "total": round(sum(ph.values()) / n, 4), "rho_sum": float(st["density"].astype(np.float64).sum()),
                      "v_abs": float(np.abs(st["vel"].astype(np.float64)).sum())}))
else:
    for lib in sys.argv[1:]:
        env = dict(os.environ, SPH_HIP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True, timeout=300)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + lib + " " + r.stderr[-400:]), flush=True)
