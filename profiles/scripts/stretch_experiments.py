#!/usr/bin/env python3
"""WHAT in the data makes the chip run the pair kernels at half speed for long stretches of a run?  (VERDICT r3 item 1a:
one rank's eighth of C3 -- a 256 x 256 x 32 lattice -- takes 0.96 ms per step over steps 5000..6000 and 0.64 ms over
6020..6220; round 3 found the same effect at C3 in steps 1710..1760 and ruled out cells, candidates and instruction
counts.)

Part A steps the thin lattice as a whole-domain context in blocks of 25 steps and prints ms/step, shader clock and
board power (hwmon) per block; the first block of a stretch that runs >= 1.5x the fastest block so far is kept as
S_slow (positions and velocities by creation index), a block of the first 1000 steps as S_fast.
Part B uploads those states into a fresh context and times, in blocks: the fused step as it is; the step with the
velocities zeroed; with the velocities of the other state; the DENSITY pass alone, repeated (it reads positions only);
the density pass on positions with 1e-6 of noise; after a 3 s pause.

    python profiles/scripts/stretch_experiments.py [total_steps] [lattice nx,ny,nz]     (GPU box, repo root)
"""
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from gpufluidsimulator_amd import capi, ic  # noqa: E402

DT = float(ic.DEFAULT_DT)


def sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "power1_input", "power1_average"):
            try:
                out[name] = int(open(os.path.join(hw, name)).read())
            except Exception:
                pass
    return out


def sens_str(s):
    return f"sclk {s.get('freq1_input', 0) / 1e6:6.0f} MHz  power {s.get('power1_input', s.get('power1_average', 0)) / 1e6:6.0f} W"


def timed_steps(c, k):
    t0 = time.perf_counter()
    c.step(DT, k)
    mid = sensors()
    c.sync()
    return (time.perf_counter() - t0) / k * 1e3, mid


def timed_density(c, k):
    t0 = time.perf_counter()
    for _ in range(k):
        c.density()
    mid = sensors()
    c.sync()
    return (time.perf_counter() - t0) / k * 1e3, mid


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 6200
    lattice = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (256, 256, 32)
    cfg = ic.CONFIGS["C3"]
    n = lattice[0] * lattice[1] * lattice[2]
    blk = 25
    states = {}
    print(f"== part A: {lattice} = {n} particles, whole-domain context, blocks of {blk} steps", flush=True)
    with capi.Context(n, box=cfg["box"], grid=cfg["grid"]) as c:
        c.reset_lattice(lattice, jitter=True)
        c.step(DT, 5); c.sync()
        best, slow_run, s = 1e9, 0, 5
        while s < total:
            ms, mid = timed_steps(c, blk)
            s += blk
            best = min(best, ms)
            slow = ms > 1.5 * best
            slow_run = slow_run + 1 if slow else 0
            print(f"steps {s - blk:5d}..{s:5d}: {ms:7.3f} ms/step  {sens_str(mid)}{'   SLOW' if slow else ''}", flush=True)
            if s >= 800 and "fast" not in states and not slow:
                states["fast"] = (s, c.download(want=("pos", "vel")))
            if slow_run == 4 and "slow" not in states:
                states["slow"] = (s, c.download(want=("pos", "vel")))
            if slow_run == 4 and "slow" in states and s > states["slow"][0] + 1500 and "slow2" not in states:
                states["slow2"] = (s, c.download(want=("pos", "vel")))
    if "slow" not in states:
        print("no slow stretch found"); return
    print({k: v[0] for k, v in states.items()}, flush=True)

    def fresh(pos, vel):
        c = capi.Context(n, box=cfg["box"], grid=cfg["grid"])
        c.upload(np.ascontiguousarray(pos), np.ascontiguousarray(vel))
        return c

    def run_steps(label, pos, vel, pause=0.0, blocks=6):
        c = fresh(pos, vel)
        c.step(DT, 2); c.sync()
        if pause:
            time.sleep(pause)
        out = []
        for _ in range(blocks):
            ms, mid = timed_steps(c, blk)
            out.append(f"{ms:6.3f}")
        print(f"  {label:58s} ms/step per block of {blk}: {' '.join(out)}   {sens_str(mid)}", flush=True)
        c.close()

    def run_density(label, pos, vel, blocks=6):
        c = fresh(pos, vel)
        c.hash(); c.sort(); c.build_cells(); c.density(); c.sync()
        out = []
        for _ in range(blocks):
            ms, mid = timed_density(c, 4 * blk)
            out.append(f"{ms:6.3f}")
        print(f"  {label:58s} ms/launch per block of {4 * blk}: {' '.join(out)}   {sens_str(mid)}", flush=True)
        c.close()

    rng = np.random.default_rng(3)
    f, sl = states["fast"][1], states["slow"][1]
    print("== part B: states re-uploaded into fresh contexts", flush=True)
    for rep in range(2):
        run_steps(f"fused step, S_fast (step {states['fast'][0]})", f["pos"], f["vel"])
        run_steps(f"fused step, S_slow (step {states['slow'][0]})", sl["pos"], sl["vel"])
    run_steps("fused step, S_slow after a 3 s pause", sl["pos"], sl["vel"], pause=3.0)
    run_steps("fused step, S_slow positions, velocities = 0", sl["pos"], np.zeros_like(sl["vel"]))
    run_steps("fused step, S_slow positions, S_fast velocities", sl["pos"], f["vel"])
    run_steps("fused step, S_fast positions, S_slow velocities", f["pos"], sl["vel"])
    run_density("density alone, S_fast", f["pos"], f["vel"])
    run_density("density alone, S_slow", sl["pos"], sl["vel"])
    noise = rng.uniform(-1e-6, 1e-6, sl["pos"].shape).astype(np.float32)
    run_density("density alone, S_slow + 1e-6 noise", sl["pos"] + noise, sl["vel"])
    noise = rng.uniform(-1e-4, 1e-4, sl["pos"].shape).astype(np.float32)
    run_density("density alone, S_slow + 1e-4 noise", sl["pos"] + noise, sl["vel"])
    run_density("density alone, S_slow shifted by (0.01, 0.01, 0.01)", sl["pos"] + np.float32(0.01), sl["vel"])
    if "slow2" in states:
        s2 = states["slow2"][1]
        run_steps(f"fused step, S_slow2 (step {states['slow2'][0]})", s2["pos"], s2["vel"])
        run_density("density alone, S_slow2", s2["pos"], s2["vel"])


if __name__ == "__main__":
    main()
