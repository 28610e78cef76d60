"""Long C3 run: when does the dam start to flow, how many particles change cell per step, does it stay finite."""
import json
import sys
import time

import numpy as np
import torch  # noqa: F401  (HIP runtime first)

sys.path.insert(0, ".")
from gpufluidsimulator_amd import capi, ic

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 100
cfg = ic.CONFIGS[wl]
n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
ctx = capi.Context(n, box=cfg["box"], grid=cfg["grid"], device=0)
ctx.reset_lattice(cfg["lattice"], jitter=True)
dt = float(ic.DEFAULT_DT)
done = 0
prev = ctx.sort_stats()
k_it = 0
while done < total:
    k_it += 1
    t0 = time.perf_counter()
    ctx.step(dt, chunk)
    ctx.sync()
    ms = (time.perf_counter() - t0) / chunk * 1e3
    done += chunk
    st = ctx.sort_stats()
    rec = {"step": done, "ms_per_step": round(ms, 3), "last_movers": st["last_movers"],
           "skips": st["skips"] - prev["skips"], "merges": st["merges"] - prev["merges"],
           "sorts": st["sorts"] - prev["sorts"], "movers_mean": (st["movers_total"] - prev["movers_total"]) / chunk}
    prev = st
    if k_it % 10 == 0:
        s = ctx.download(want=("vel", "density"))
        v = np.linalg.norm(s["vel"], axis=1)
        rec.update(vmax=float(v.max()), vmean=float(v.mean()), rho_max=float(s["density"].max()),
                   rho_mean=float(s["density"].mean()), finite=bool(np.isfinite(s["vel"]).all()))
        ctx.timing(True); ctx.timing_reset(); ctx.step(dt, 10); ctx.sync()
        ph, k = ctx.timing_get(); ctx.timing(False)
        done += 10
        prev = ctx.sort_stats()
        rec["phases_ms"] = {a: round(b / k, 3) for a, b in ph.items()}
    print(json.dumps(rec), flush=True)
