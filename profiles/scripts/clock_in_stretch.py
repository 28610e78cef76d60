#!/usr/bin/env python3
"""Shader clock and board power while C3 runs through the slow stretch of its run-up (steps ~1700-1760): chunks of 10
steps, wall time per step, and after every chunk the GPU's sysfs hwmon readings (freq1_input = sclk, power1_average /
power1_input) -- to tell a clock effect from a work effect.
    python profiles/scripts/clock_in_stretch.py [first_step last_step]"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from gpufluidsimulator_amd import capi, ic  # noqa: E402


def sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input"):
            p = os.path.join(hw, name)
            try:
                out[os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(hw)))) + ":" + name] = int(open(p).read())
            except Exception:
                pass
    return out


lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1500, 1900)
cfg = ic.CONFIGS["C3"]
n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
dt = float(ic.DEFAULT_DT)
print("sensors found:", sorted(sensors()))
with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
    c.reset_lattice(cfg["lattice"], jitter=True)
    c.step(dt, lo); c.sync()
    s = lo
    while s < hi:
        t0 = time.perf_counter()
        c.step(dt, 10)
        mid = sensors()                      # sampled while the 10 steps are in flight
        c.sync()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        s += 10
        print(f"steps {s - 10:5d}..{s:5d}: {ms:6.3f} ms/step  " + "  ".join(f"{k.split(':')[1]} {v}" for k, v in sorted(mid.items())), flush=True)
