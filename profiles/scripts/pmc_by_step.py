#!/usr/bin/env python3
"""Counter values of one kernel by dispatch ordinal (= step) from a rocprofv3 --pmc counter_collection CSV: means over
windows of steps, to compare a slow stretch of the run with the stretch before it.
    python profiles/scripts/pmc_by_step.py <counter_collection.csv> <kernel-substring> <lo:hi> [<lo:hi> ...]"""
import csv
import sys
from collections import defaultdict

path, pat = sys.argv[1], sys.argv[2]
per = defaultdict(dict)
for r in csv.DictReader(open(path, newline="")):
    if pat in r["Kernel_Name"]:
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(per)
for w in sys.argv[3:]:
    lo, hi = (int(v) for v in w.split(":"))
    sel = ids[lo:hi]
    names = sorted(per[sel[0]])
    print(f"{pat} steps {lo}..{hi}: " + ", ".join(f"{c} {sum(per[d][c] for d in sel) / len(sel):.4g}" for c in names))
