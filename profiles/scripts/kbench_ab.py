#!/usr/bin/env python3
"""A/B of library variants in the flowing C3 state with REPETITIONS: kbench_flow.py's protocol (one process per run, the
same snapshot, 10 + 40 steps with the device timers on), every variant R times in interleaved order; prints the
median and the minimum of the per-phase means.  Single runs differ by +-3 % on one box (the chip sits at its power
cap), so one run per variant cannot resolve a 1-2 % change.
    python profiles/scripts/kbench_ab.py R scratch/v/libsph_A.so scratch/v/libsph_B.so ...   (GPU box, repo root)"""
import json
import os
import statistics
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
R = int(sys.argv[1])
libs = sys.argv[2:]
snap = os.environ.get("KB_SNAP", "/tmp/c3_flow.snap")
if not os.path.exists(snap):
    subprocess.run([sys.executable, os.path.join(HERE, "kbench_flow.py"), "--one", "prepare"], check=True, timeout=900)
res = {l: [] for l in libs}
# a "library" may carry environment settings: path.so@SPH_BLOCK_ORDER=0,0,4@OTHER=1
def split(spec):
    parts = spec.split("@")
    return parts[0], dict(p.split("=", 1) for p in parts[1:])


for rep in range(R):
    for lib in libs:
        path, extra = split(lib)
        env = dict(os.environ, SPH_HIP_LIB=os.path.abspath(path), **extra)
        r = subprocess.run([sys.executable, os.path.join(HERE, "kbench_flow.py"), "--one", "run"], env=env, capture_output=True,
                           text=True, timeout=600)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
        try:
            res[lib].append(json.loads(line))
        except Exception:
            print("FAILED", lib, r.stderr[-400:], flush=True)
for lib in libs:
    rows = res[lib]
    if not rows:
        continue
    out = {"lib": os.path.basename(lib), "runs": len(rows)}
    for k in ("ms_per_step", "dens", "force", "sort"):
        v = [r[k] for r in rows if k in r]
        out[k] = {"median": round(statistics.median(v), 4), "min": round(min(v), 4)}
    out["rho_sum"] = rows[0]["rho_sum"]; out["v_abs"] = rows[0]["v_abs"]
    print(json.dumps(out), flush=True)
