#!/usr/bin/env python3
"""Duration of every dispatch of one kernel, in launch order, from a rocprofv3 --kernel-trace CSV -- to find WHICH steps
the slow launches belong to (the --stats table only gives min / max / average).

    python profiles/kernel_series.py <kernel_trace.csv> <kernel-substring> [out.json]

Prints the 12 longest dispatches with their ordinal (= time step for a kernel launched once per step) and the mean of
their neighbours, and writes the whole series (microseconds, one number per dispatch) as JSON."""
import csv
import json
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if pat in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    dur = [(e - s) / 1e3 for s, e in rows]
    order = sorted(range(len(dur)), key=lambda i: -dur[i])[:12]
    print(f"{pat}: {len(dur)} dispatches, mean {sum(dur) / len(dur):.1f} us, max {max(dur):.1f} us")
    for i in sorted(order):
        lo, hi = max(0, i - 5), min(len(dur), i + 6)
        nb = [dur[k] for k in range(lo, hi) if k != i]
        print(f"  dispatch {i:6d}: {dur[i]:8.1f} us   (neighbours +-5: mean {sum(nb) / len(nb):8.1f} us)")
    if len(sys.argv) > 3:
        json.dump({"kernel": pat, "us": [round(d, 1) for d in dur]}, open(sys.argv[3], "w"))


if __name__ == "__main__":
    main()
