#!/usr/bin/env python3
"""Per-kernel PMC means over the TIMED WINDOW of bench.py (the dispatches between the (runup+warmup)-th and the
(runup+warmup+steps)-th k_force<1,1,1> launch), from the counter_collection CSVs of profiles/collect_pmc.sh.

Writes profiles/pmc_traffic.json (HBM-side traffic per launch: FETCH_SIZE x 2 per the gfx950 correction of
MI355X_MICROARCH.md section HBM -- FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced reads -- plus
WRITE_SIZE, both reported in KB) and profiles/<tag>_c3_flow_sq_counters.json.

    python profiles/pmc_summarize.py <dir with fetch/ write/ sq1/ sq2/> <runup> <warmup> <steps>
"""
import csv, glob, json, os, sys
from collections import defaultdict

root, runup, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tag = sys.argv[5] if len(sys.argv) > 5 else "r05"
HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    return name.split("(")[0].replace("void ", "").replace("sph::", "").strip()


def window_means(path):
    rows = list(csv.DictReader(open(path, newline="")))
    per_dispatch = defaultdict(dict)          # dispatch id -> {counter: value}, kernel name
    names = {}
    for r in rows:
        d = int(r["Dispatch_Id"])
        per_dispatch[d][r["Counter_Name"]] = per_dispatch[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[d] = short(r["Kernel_Name"])
    ids = sorted(per_dispatch)
    marks = [d for d in ids if names[d].startswith("k_force<true, true, true") or names[d].startswith("k_force<1, 1, 1>")]
    lo, hi = marks[runup + warm - 1], marks[runup + warm + steps - 1]
    acc = defaultdict(lambda: defaultdict(list))
    for d in ids:
        if lo < d <= hi:
            for k, v in per_dispatch[d].items():
                acc[names[d]][k].append(v)
    return {n: {k: sum(v) / len(v) for k, v in c.items()} | {"_launches_per_step": len(next(iter(c.values()))) / steps}
            for n, c in acc.items()}


res = {}
for p in ("fetch", "write", "sq1", "sq2"):
    f = glob.glob(os.path.join(root, p, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        print("missing pass", p); continue
    for n, c in window_means(f[0]).items():
        res.setdefault(n, {}).update(c)
N = 16777216
force = next((v for k, v in res.items() if k.startswith("k_force<")), {})
dens = next((v for k, v in res.items() if k.startswith("k_density") and not k.startswith("k_density_h")), {})
traffic = {
    "workload": "C3", "state": "flow", "round": int(tag.lstrip("r") or 0), "collected": __import__("time").strftime("%Y-%m-%d"),
    "method": f"profiles/collect_pmc.sh: rocprofv3 --pmc, one pass per counter group, over `python bench.py --runup {runup} "
              f"--steps {steps} --warmup {warm} --no-cpu`; means over the dispatches of the timed window; FETCH_SIZE and WRITE_SIZE in "
              "KB; reads = FETCH_SIZE x 2 (gfx950: wide coalesced reads are tallied at half their bytes, MI355X_MICROARCH.md HBM); "
              "both count fabric requests of the L2s (Infinity-Cache hits included): an upper bound on HBM traffic",
    "raw": {k: {c: v.get(c) for c in ("FETCH_SIZE", "WRITE_SIZE", "_launches_per_step") if c in v} for k, v in res.items()},
}
if "FETCH_SIZE" in force and "WRITE_SIZE" in force:
    traffic["force_fused_hbm_bytes_per_launch"] = int((2 * force["FETCH_SIZE"] + force["WRITE_SIZE"]) * 1024)
    traffic["force_fused_algorithmic_bytes_per_launch"] = 84 * N
if "FETCH_SIZE" in dens and "WRITE_SIZE" in dens:
    traffic["density_hbm_bytes_per_launch"] = int((2 * dens["FETCH_SIZE"] + dens["WRITE_SIZE"]) * 1024)
    traffic["density_algorithmic_bytes_per_launch"] = 20 * N
json.dump(traffic, open(os.path.join(HERE, "pmc_traffic.json"), "w"), indent=1)
sq = {k: {c: v[c] for c in v if c.startswith("SQ_") or c.startswith("GRBM") or c.startswith("_")} for k, v in res.items()}
for k, v in sq.items():                         # per-wave instruction counts (what the kernels are bound by)
    if v.get("SQ_WAVES"):
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
            if c in v:
                v[c + "_per_wave"] = v[c] / v["SQ_WAVES"]
json.dump(sq, open(os.path.join(HERE, tag + "_c3_flow_sq_counters.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1)[:3000])
print(json.dumps({k: sq[k] for k in sq if k.startswith("k_force<") or k.startswith("k_density")}, indent=1))
