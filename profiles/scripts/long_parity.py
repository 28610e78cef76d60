#!/usr/bin/env python3
"""How far a FREE run of the HIP library drifts from the CPU oracle over hundreds of steps, for several library builds
(the integrate epilogue takes v_rcp_f32 where the reference divides -- is the claimed tolerance measured?).
    python profiles/scripts/long_parity.py <fixture> <steps> lib1.so lib2.so ...      (GPU box, repo root)
Every `every` steps: max |dpos| / box, max |dvel| / |v|max, max |drho / rho|, particles beyond 1e-5 |v|max, and the number
of particles whose collision COUNT of that step differs (the discrete event free runs part at)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.getcwd()
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch  # noqa: F401
    from conftest import load_golden
    from gpufluidsimulator_amd import capi
    from oracle import oracle
    name, steps, every = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    g = load_golden(name)
    dt, box = float(g["dt"]), float(g["box"].max())
    o = oracle.Oracle(g["pos"], g["vel"], g["box"], g["grid"], oracle.CELL_LINEAR)
    rows = []
    with capi.Context(g["pos"].shape[0], box=g["box"], grid=g["grid"]) as c:
        c.upload(g["pos"], g["vel"])
        for k in range(0, steps, every):
            c.step(dt, every); o.step(dt, every)
            a, b = c.download(), o.state()
            ev = np.abs(a["vel"] - b["vel"]).max(axis=1) / np.abs(b["vel"]).max()
            rows.append({"step": k + every, "dpos": float(np.abs(a["pos"] - b["pos"]).max() / box), "dvel": float(ev.max()),
                         "drho": float(np.abs(a["density"] / b["density"] - 1).max()), "beyond_1e-5": int((ev > 1e-5).sum()),
                         "beyond_1e-4": int((ev > 1e-4).sum())})
    o.close()
    print(json.dumps(rows))
    sys.exit(0)

name, steps = sys.argv[1], int(sys.argv[2])
every = max(steps // 12, 1)
for lib in sys.argv[3:]:
    env = dict(os.environ, SPH_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, __file__, "--one", name, str(steps), str(every)], env=env, capture_output=True, text=True, timeout=1200)
    print(f"== {os.path.basename(lib)}  fixture {name}, free run against the oracle ==", flush=True)
    try:
        for row in json.loads(r.stdout.strip().splitlines()[-1]):
            print("  step %4d  dpos/box %.2e  dvel/|v|max %.2e  drho/rho %.2e  particles beyond 1e-5: %d, beyond 1e-4: %d" %
                  (row["step"], row["dpos"], row["dvel"], row["drho"], row["beyond_1e-5"], row["beyond_1e-4"]), flush=True)
    except Exception:
        print("FAILED", r.stderr[-800:], flush=True)
