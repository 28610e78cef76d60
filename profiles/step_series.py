#!/usr/bin/env python3
"""Per-STEP series of a rocprofv3 --kernel-trace CSV: span, busy time, idle time and the time of every kernel for each
step, and a comparison of step ranges -- written to explain why a long stretch of steps (the "sustained" figure of
bench.py) is slower than a short window of the same run.

    python profiles/step_series.py <kernel_trace.csv> <marker-substring> <out.json> <lo:hi> [<lo:hi> ...]

A step starts at a dispatch of the marker kernel (one per step: `k_slab_bounds_pack` is NOT first in a slab step, so
the slab step uses `k_mm_compact` / `k_hash` ...: pass the kernel that opens the step's sort, e.g. k_mm_compact, and
the span of step k is marker[k] .. marker[k+1]).  For every range lo:hi (step numbers, hi exclusive) the script
prints mean span / busy / idle and the mean time per kernel, the class of every step by the sort that ran in it
(`small`: the one-block sort took the movers, `passes`: the multi-block radix passes did, `full`: full radix sort +
gather), and the distribution of spans.  The JSON holds the per-step table (microseconds)."""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0].replace("void ", "").replace("sph::", "").replace("(anonymous namespace)::", "")
    return name.strip()


def main():
    path, marker, out = sys.argv[1], sys.argv[2], sys.argv[3]
    ranges = [tuple(int(v) for v in a.split(":")) for a in sys.argv[4:]]
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    steps = []
    for k in range(len(marks) - 1):
        win = rows[marks[k]:marks[k + 1]]
        t0, t1 = win[0][0], rows[marks[k + 1]][0]
        busy, cur = 0, t0
        per = defaultdict(float)
        for s, e, name in win:
            per[name] += (e - s) / 1e3
            if s > cur:
                busy += e - s; cur = e
            elif e > cur:
                busy += e - cur; cur = e
        small = per.get("k_os_small<8>", 0.0) + per.get("k_os_small<9>", 0.0)
        passes = sum(v for n, v in per.items() if n.startswith("k_os_pass"))
        npass = sum(1 for _, _, n in win if n.startswith("k_os_pass"))
        cls = "full" if "k_reorder" in per else ("passes" if npass and passes > 9.0 * npass else ("small" if small > 9.0 else "none"))
        steps.append({"step": k, "span": (t1 - t0) / 1e3, "busy": busy / 1e3, "class": cls, "kernels": dict(per)})
    json.dump({"marker": marker, "steps": [{"step": s["step"], "span": round(s["span"], 1), "busy": round(s["busy"], 1),
                                           "class": s["class"],
                                           "kernels": {k: round(v, 1) for k, v in s["kernels"].items()}} for s in steps]},
              open(out, "w"))
    print(f"{len(steps)} steps in the trace (marker {marker})")
    for lo, hi in ranges:
        sel = [s for s in steps if lo <= s["step"] < hi]
        if not sel:
            print(f"range {lo}:{hi}: no steps"); continue
        n = len(sel)
        span = sum(s["span"] for s in sel) / n
        busy = sum(s["busy"] for s in sel) / n
        print(f"\n== steps {lo}..{hi - 1} ({n}): span {span:.1f} us, busy {busy:.1f} us, idle {span - busy:.1f} us per step")
        by_cls = defaultdict(list)
        for s in sel:
            by_cls[s["class"]].append(s["span"])
        for c, v in sorted(by_cls.items()):
            print(f"   sort class {c:7s}: {len(v):5d} steps ({100.0 * len(v) / n:5.1f} %), mean span {sum(v) / len(v):8.1f} us, "
                  f"max {max(v):8.1f} us, share of the range's time {100.0 * sum(v) / (span * n):5.1f} %")
        spans = sorted(s["span"] for s in sel)
        q = lambda p: spans[min(n - 1, int(p * n))]
        print(f"   span quantiles: 10 % {q(.1):.0f}  50 % {q(.5):.0f}  90 % {q(.9):.0f}  99 % {q(.99):.0f}  max {spans[-1]:.0f} us")
        per = defaultdict(float)
        for s in sel:
            for k, v in s["kernels"].items():
                per[k] += v
        for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:16]:
            print(f"   {k:36s} {v / n:8.1f} us per step")


if __name__ == "__main__":
    main()
