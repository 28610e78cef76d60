#!/usr/bin/env python3
"""Per-kernel statistics of ONE window of a rocprofv3 --kernel-trace run of bench.py.

bench.py steps the dam thousands of times before its timed window (state preparation), so rocprofv3's own
--stats table averages over regimes that are not the measured one.  This script reads the kernel-trace CSV and
keeps only the dispatches between the (skip+1)-th and the (skip+count)-th launch of the step's last kernel
(k_force<true,true,true>, one per time step): skip = run-up + warm-up steps, count = timed steps.

    python profiles/trace_window.py <kernel_trace.csv> <skip_steps> <count_steps> [out.csv]
"""
import csv
import sys
from collections import defaultdict


def short(name):
    # "void sph::k_force<true, true, true>(float4 const*, ...)" -> "k_force<true, true, true>"
    name = name.split("(")[0].replace("void ", "").replace("sph::", "").replace("(anonymous namespace)::", "")
    return name.strip()


def main():
    path, skip, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else None
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith("k_force<true, true, true") or r[2].startswith("k_force<1, 1, 1>")]
    if len(marks) < skip + count:
        sys.exit(f"only {len(marks)} steps in the trace, need {skip + count}")
    first = marks[skip - 1] + 1 if skip > 0 else 0          # first dispatch after the last skipped step
    last = marks[skip + count - 1]                          # the window's last k_force
    t0, t1 = rows[first][0], rows[last][1]
    agg = defaultdict(lambda: [0, 0, 10**18, 0])
    for s, e, k in rows[first:last + 1]:
        a = agg[k]
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    busy = sum(a[1] for a in agg.values())
    lines = ["kernel,calls,calls_per_step,total_ns,avg_ns,min_ns,max_ns,percent_of_busy"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{k}\",{a[0]},{a[0] / count:.2f},{a[1]},{a[1] / a[0]:.0f},{a[2]},{a[3]},{100.0 * a[1] / busy:.2f}")
    lines.append(f"\"#window: steps {skip + 1}..{skip + count}; span {t1 - t0} ns = {(t1 - t0) / count / 1e6:.4f} ms per step; "
                 f"kernel-busy {busy} ns = {busy / count / 1e6:.4f} ms per step\",,,,,,,")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
