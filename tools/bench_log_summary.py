#!/usr/bin/env python3
"""Summarise an SPH benchmark log: mean of every field over the sample lines, and each phase as % of total.

Row f4 of SURVEY.md section 8.  The reference ships a parser for its logs (`/root/reference/benchmark.py:4-67`): it
skips the two header lines, averages every field of every sample line and prints each phase as a percentage of
`total`.  Its regex (`benchmark.py:12,13`) reads only the line form of the 18 logs committed under
`benchmarks/oscar/` -- `2.005sec<TAB>total:..ns,<TAB><TAB>copying:..ns, ... FPS:189.322fps` -- and neither the form the
reference's CURRENT source writes (`SPH/particleSystem.cpp:697-716`: `<int>sec ... frames:<n>frames`) nor a
`FPS:0fps` without a decimal point (its own sequential logs).  This tool has the same semantics and reads all of them:

    <seconds>sec  total:<ns>ns,  [copying: z-index: sort: b-grid: b'-grid:]  dens: force: collision: integrate:  FPS:<x>fps | frames:<n>frames

so a log written by this build (`sph_headless -log=f [-logstyle=oscar]`, `ParticleSystem::setBenchmarkLog`) and a log
the reference published are summarised by one tool and can be laid side by side:

    python tools/bench_log_summary.py benchmarks/oscar/131072/benchmark_CUDA_*.txt  my_run.txt
    python tools/bench_log_summary.py --json my_run.txt

Fields are nanoseconds per update as the log holds them; the table prints microseconds.
"""
from __future__ import annotations

import json
import re
import sys

PHASES = ("copying", "z-index", "sort", "b-grid", "b'-grid", "dens", "force", "collision", "integrate")
FIELDS = ("total",) + PHASES

_NUM = r"([-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?)"
_HEAD = re.compile(r"^\s*" + _NUM + r"sec\b")
_FIELD = re.compile(r"([a-z][a-z'\-]*):\s*(\d+)ns")
_TAIL = re.compile(r"(FPS|frames):\s*" + _NUM + r"(fps|frames)\s*$")


def parse_line(line: str):
    """One sample line -> {"sec": float, "total": ns, <phase>: ns ..., "fps" | "frames": float}, or None if it is not one."""
    line = line.strip()
    head = _HEAD.match(line)
    tail = _TAIL.search(line)
    if not head or not tail:
        return None
    if (tail.group(1) == "FPS") != (tail.group(3) == "fps"):
        return None
    out = {"sec": float(head.group(1))}
    for name, ns in _FIELD.findall(line):
        if name in FIELDS:
            out[name] = float(ns)
    if "total" not in out or "dens" not in out:
        return None
    out["fps" if tail.group(1) == "FPS" else "frames"] = float(tail.group(2))
    return out


def summarise(path: str) -> dict:
    """Means over all sample lines (the two header lines are skipped, as `benchmark.py:6-7` does) and % of total."""
    with open(path) as f:
        title = f.readline().strip()
        mode = f.readline().strip()
        rows = [r for r in (parse_line(x) for x in f) if r]
    if not rows:
        raise ValueError(f"{path}: no sample line in either the 'FPS:..fps' or the 'frames:..frames' form")
    keys = [k for k in FIELDS + ("fps", "frames") if any(k in r for r in rows)]
    # a field that a line does not carry counts as 0 for that line, as in benchmark.py (its second pattern leaves the
    # grid-path fields of a sequential log at 0)
    mean = {k: sum(r.get(k, 0.0) for r in rows) / len(rows) for k in keys}
    pct = {k: 100.0 * mean[k] / mean["total"] for k in PHASES if k in mean and mean["total"] > 0}
    return {"file": path, "title": title, "mode": mode.split(":", 1)[-1].strip(), "samples": len(rows),
            "seconds": rows[-1]["sec"], "style": "oscar" if "fps" in mean else "frames", "mean_ns": mean,
            "percent_of_total": pct}


def _table(s: dict) -> str:
    m, p = s["mean_ns"], s["percent_of_total"]
    lines = [f"{s['file']}: {s['mode']}, {s['samples']} samples over {s['seconds']:g} s ({s['style']} form)",
             f"  {'total':<10} {m['total'] / 1e3:14.1f} us per update"]
    for k in PHASES:
        if k in m:
            lines.append(f"  {k:<10} {m[k] / 1e3:14.1f} us {p.get(k, 0.0):7.2f} %")
    if "fps" in m:
        lines.append(f"  {'fps':<10} {m['fps']:14.2f}")
    if "frames" in m:
        lines.append(f"  {'frames':<10} {m['frames']:14.1f}  (a frame count, not a rate: SPH/particleSystem.cpp:713)")
    return "\n".join(lines)


def main(argv) -> int:
    as_json = "--json" in argv
    files = [a for a in argv if not a.startswith("--")]
    if not files:
        print(__doc__)
        return 2
    out = [summarise(f) for f in files]
    if as_json:
        print(json.dumps(out if len(out) > 1 else out[0]))
    else:
        print("\n\n".join(_table(s) for s in out))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
