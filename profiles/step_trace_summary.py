#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 --kernel-trace CSV: kernel time by kernel, span per step and WHERE THE DEVICE IDLES.

    python profiles/step_trace_summary.py <kernel_trace.csv> <marker-substring> <skip_steps> <count_steps> [out.csv]

A step starts at a dispatch of the marker kernel (exactly one per step: `k_slab_bounds_pack` for the slab step,
`k_mm_compact` for the whole-domain step with the merge sort).  Steps skip+1 .. skip+count are kept.  Besides the
per-kernel table the script lists the idle gaps (device has nothing running) by the kernel that ENDS the gap, i.e.
the launch the device was waiting for -- the fixed costs of a small step (host wait, event hops, launch latency).
Kernels on different streams overlap; busy time is the union of the dispatch intervals.
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0].replace("void ", "").replace("sph::", "").replace("(anonymous namespace)::", "")
    return name.strip()


def main():
    path, marker, skip, count = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    out = sys.argv[5] if len(sys.argv) > 5 else None
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < skip + count + 1:
        sys.exit(f"only {len(marks)} steps in the trace, need {skip + count + 1}")
    first, last = marks[skip], marks[skip + count]          # [first, last): count whole steps
    win = rows[first:last]
    t0, t1 = win[0][0], rows[last][0]
    agg = defaultdict(lambda: [0, 0, 10**18, 0])
    for s, e, k in win:
        a = agg[k]
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    # union of the intervals and the gaps between them
    gaps = defaultdict(lambda: [0, 0])
    busy, cur_end = 0, t0
    for s, e, k in win:
        if s > cur_end:
            g = gaps[k]
            g[0] += 1; g[1] += s - cur_end
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    if t1 > cur_end:
        g = gaps["<next step's first kernel>"]
        g[0] += 1; g[1] += t1 - cur_end
    total = sum(a[1] for a in agg.values())
    # per kernel: the UNION of its dispatch intervals (launches of one kernel on two streams overlap: their times add up to more
    # than the time during which that kernel was running at all -- k_density's two launches of a slab step, k_force's interior
    # and boundary launches)
    union = defaultdict(int)
    ends = {}
    for s, e, k in win:
        ce = ends.get(k, 0)
        if s >= ce:
            union[k] += e - s; ends[k] = e
        elif e > ce:
            union[k] += e - ce; ends[k] = e
    lines = ["kernel,calls,calls_per_step,total_ns,avg_ns,min_ns,max_ns,ns_per_step,union_ns_per_step"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{k}\",{a[0]},{a[0] / count:.2f},{a[1]},{a[1] / a[0]:.0f},{a[2]},{a[3]},{a[1] / count:.0f},{union[k] / count:.0f}")
    lines.append(f"\"#window: steps {skip + 1}..{skip + count}; span {(t1 - t0) / count / 1e3:.1f} us per step; device busy (union) "
                 f"{busy / count / 1e3:.1f} us per step; idle {(t1 - t0 - busy) / count / 1e3:.1f} us per step; sum of kernel "
                 f"times {total / count / 1e3:.1f} us per step; {len(win) / count:.1f} dispatches per step\",,,,,,,")
    lines.append("\"#idle gaps by the kernel that ends them: kernel, gaps per step, us per step\",,,,,,,")
    for k, g in sorted(gaps.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"#gap before {k}\",{g[0] / count:.2f},{g[1] / count / 1e3:.2f},,,,,")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
