#!/usr/bin/env python3
"""bench.py -- particle-steps/sec of the SPH step on MI355X (see BASELINE.json / DESIGN.md section 5).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--runup R] [--workload C3|C2|C1|C4|C5] [--no-cpu]

N = 1: the whole-domain context runs BASELINE config 3 (dam-break, 16,777,216 particles, 512^3 grid, dt 5e-7)
with the state resident in HBM.  The timed state is a FLOWING dam: the same initial lattice is first stepped R
times on the device (state preparation, not timed; at dt = 5e-7 nothing crosses a cell face for the first ~1000
steps), so that in the timed window every phase of the reference's step does work -- cell hash, sort, cell table,
density, force, collision, integrate (SPH/particleSystem.cpp:773-795).  Then W warm-up steps and K timed steps
between two device syncs.  `value` is that figure; `value_at_rest` (fresh lattice) and `value_full_sort` (the same
flowing state with the radix sort forced every step) are reported next to it.
N > 1 (launched by torch.distributed.run, one rank per GPU): z-slabs, see gpufluidsimulator_amd/slab.py.  The default is
STRONG scaling of the same config 3 (BASELINE.json's metric: "dam-break 16M particles, 1/2/4/8 MI355X"); `--scaling weak`
gives every GPU its own 16.7 M particles, `--workload C4` is BASELINE config 4.  `--gpus N --one-gpu --transport local`
rehearses the N-rank step as N threads on one GPU.

One JSON line on stdout (rank 0).  `roofline` prices the dominant kernel (the fused force+collision+integrate
traversal) by its ALGORITHMIC bytes over its mean device time measured with HIP events on the library's stream
during the flowing steps; `cpu_baseline` times the reference's own OpenMP code (oracle/_ref, kind "reference") and
the C restatement (kind "port") on the host cores, on a bounded sample (a 64^3-particle dam break).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gpufluidsimulator_amd import capi, ic  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak
# algorithmic (compulsory) bytes per particle per launch, sorted-SoA, every array streamed once
# 
BYTES_PER_PARTICLE = {
    "dens": 20,                 # R pos 12 ; W rho 4 + p 4
    # R pos 12 + vel 12 + rho 4 + p 4 + index 4 = 36 ; W pos 12 + vel 12 + index 4 + gl_pos float4 16 + next key 4 = 48
    "force_fused": 36 + 48,
    # merge path of the sort: R old key 4 + mover mask ; gather R (key,slot) 8 + pos/vel 32, W pos/vel 32 + key 4
    "sort_merge": 80,
}
# useful flops: candidates x per-pair arithmetic of the reference formulas
FLOP_PER_PARTICLE = {"dens": 216 * 11, "force_fused": 216 * 34}
DEFAULT_RUNUP = {"C3": 6000, "C2": 6000, "C1": 4000, "C4": 6000, "C5": 6000}


def _dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: same step, host cores of this box, bounded sample
# ---------------------------------------------------------------------------------------------------------------
def _median_rate(run, reps=3):
    rates = [run() for _ in range(reps)]
    return statistics.median(rates), rates


def cpu_headline(cfg, pos, vel, threads_port, threads_ref, label):
    """ONE step of the C restatement and of the reference binary on the HEADLINE configuration itself (the particles
    of the GPU's timed window, by creation index): SURVEY 8d / BASELINE.md section 3, "same ICs, same step".  Team
    sizes are the best ones of the 262,144-particle probe.  Each leg is a single run (a step takes seconds)."""
    from oracle import oracle, refio
    n = pos.shape[0]
    dt = float(ic.DEFAULT_DT)
    out = {"sample": label, "particles": int(n), "steps_per_run": 1}
    o = oracle.Oracle(pos, vel, cfg["box"], cfg["grid"], oracle.CELL_MORTON, fast="native")
    o.L.orc_set_num_threads(threads_port)
    t0 = time.perf_counter(); o.step(dt, 1); sec = time.perf_counter() - t0
    o.close()
    out["port"] = {"value": n / sec, "cores": threads_port, "seconds_per_step": sec}
    if refio.available():
        _, st = refio.run_ref(pos, vel, cfg["box"], cfg["grid"][0], dt, 1, threads=threads_ref)
        out["reference"] = {"value": st["particle_steps_per_s"], "cores": threads_ref,
                            "seconds_per_step": n / max(st["particle_steps_per_s"], 1e-30)}
    kind = "reference" if "reference" in out else "port"
    out.update(value=out[kind]["value"], cores=out[kind]["cores"], kind=kind)
    return out


def cpu_baseline(budget_s=24.0, headline=None):
    """The reference's own OpenMP path (kind "reference", oracle/_ref/sph_ref: SPH/particleSystem.cpp compiled in
    the build container) and the C restatement (kind "port", oracle/sph_oracle.c compiled here with -O3
    -march=native) on a 64^3-particle dam break (BASELINE config 2 geometry): all host threads and one thread,
    median of 3 runs each -- and, `headline` = (cfg, pos, vel, label), ONE step of both on the headline configuration
    itself.  The headline `value` is the reference's figure on the headline configuration when its binary is present."""
    from oracle import oracle, refio
    cfg = ic.CONFIGS["C2"]
    pos, vel = ic.dam_break_lattice(cfg["lattice"], cfg["box"], jitter=True)
    n = pos.shape[0]
    model, cores = oracle.cpu_info()
    dt = float(ic.DEFAULT_DT)
    share = budget_s / 4.0                # four legs: {reference, port} x {all threads, one thread}
    out = {"unit": "particle-steps/s", "cpu_model": model, "host_threads": cores,
           "sample": "dam-break 64^3 = 262144 particles, 128^3 grid, dt 5e-7, jittered lattice; median of 3 runs"}
    legs = {}

    # -- port: oracle/sph_oracle.c, -O3 -march=native -fopenmp, built on this machine
    o = oracle.Oracle(pos, vel, cfg["box"], cfg["grid"], oracle.CELL_MORTON, fast="native")

    def port_run(threads, steps):
        def run():
            o.L.orc_set_num_threads(threads)
            t0 = time.perf_counter()
            o.step(dt, steps)
            return n * steps / (time.perf_counter() - t0)
        return run
    # a 262144-particle step is small for a many-core host (serial sort, one OpenMP region per phase): probe a few
    # team sizes, one step each, and keep the best -- as for the reference below
    best_t, per = 1, float("inf")
    for t in sorted({1, 8, 16, 32, 64, cores}):
        if t > cores:
            continue
        o.L.orc_set_num_threads(t)
        o.step(dt, 1)                                   # the first region of a new team pays for its threads
        t0 = time.perf_counter(); o.step(dt, 1); dtp = time.perf_counter() - t0
        if dtp < per:
            best_t, per = t, dtp
    steps_all = int(max(1, min(50, share / 3.0 / max(per, 1e-3))))
    rate, rates = _median_rate(port_run(best_t, steps_all))
    legs["port"] = {"value": rate, "cores": best_t, "steps_per_run": steps_all, "runs": rates,
                    "build": "oracle/sph_oracle.c, gcc -O3 -march=native -fopenmp (compiled on this host); team size = "
                             f"best of a probe over {{1,8,16,32,64,{cores}}} threads"}
    rate1, rates1 = _median_rate(port_run(1, 1))
    legs["port"]["one_thread"] = {"value": rate1, "runs": rates1, "steps_per_run": 1}
    o.close()

    # -- reference: the reference's z* methods, g++ -O2 -fopenmp (prebuilt where /root/reference exists)
    if refio.available():
        # its `omp parallel for schedule(static, 4)` over ALL grid cells gets slower beyond a few dozen threads:
        # probe a few team sizes (one step each) and keep the best
        best_t, best_rate = 1, 0.0
        for t in sorted({8, 16, 32, 64, cores}):
            if t > cores:
                continue
            _, probe = refio.run_ref(pos, vel, cfg["box"], cfg["grid"][0], dt, 1, threads=t)
            if probe["particle_steps_per_s"] > best_rate:
                best_t, best_rate = t, probe["particle_steps_per_s"]
        steps_all = int(max(1, min(50, share / 3.0 * best_rate / n)))

        def ref_run(threads, steps):
            def run():
                _, st = refio.run_ref(pos, vel, cfg["box"], cfg["grid"][0], dt, steps, threads=threads)
                return st["particle_steps_per_s"]
            return run
        rate, rates = _median_rate(ref_run(best_t, steps_all))
        legs["reference"] = {"value": rate, "cores": best_t, "steps_per_run": steps_all, "runs": rates,
                             "build": "SPH/particleSystem.cpp z* methods (OpenMP mode), g++ -O2 -fopenmp; team size = "
                                      f"best of a probe over {{8,16,32,64,{cores}}} threads"}
        rate1, rates1 = _median_rate(ref_run(1, 1))
        legs["reference"]["one_thread"] = {"value": rate1, "runs": rates1, "steps_per_run": 1}
    kind = "reference" if "reference" in legs else "port"
    out["reference_binary_present"] = "reference" in legs      # oracle/_ref/sph_ref travels in the snapshot (git-ignored)
    if "reference" not in legs:
        print("[bench] cpu_baseline: oracle/_ref/sph_ref is absent -- reporting the C restatement (kind 'port')", file=sys.stderr)
    out.update(value=legs[kind]["value"], cores=legs[kind]["cores"], kind=kind,
               one_thread=legs[kind]["one_thread"]["value"])
    out.update(legs)
    if headline is not None:
        # the 262,144-particle legs above stay as `small_sample`; value / cores / sample now describe the headline config
        hcfg, hpos, hvel, label = headline
        small = {k: out[k] for k in ("sample", "value", "cores", "kind", "one_thread")}
        try:
            head = cpu_headline(hcfg, hpos, hvel, legs["port"]["cores"], legs.get("reference", legs["port"])["cores"], label)
        except Exception as e:          # (out of host memory, a killed child): keep the bounded sample, say why
            out["headline_error"] = f"{type(e).__name__}: {e}"
            return out
        out["small_sample"] = small
        out["headline"] = head
        out.update(value=head["value"], cores=head["cores"], kind=head["kind"], sample=head["sample"])
    return out


# ---------------------------------------------------------------------------------------------------------------
# HBM-side traffic of the pair kernels, measured for THIS run: rocprofv3 --pmc passes over a child of this script
# ---------------------------------------------------------------------------------------------------------------
def pmc_child(args):
    """`bench.py --pmc-child <snapshot>` (started under rocprofv3 by pmc_traffic below): the flowing state the parent saved,
    W + K more steps of it.  Nothing is printed; the profiler's counter CSV is the output."""
    n, p = capi.Context.snapshot_info(args.pmc_child)
    if args.pmc_slab:       # an N > 1 run's rank 0: its slab's particles stepped ALONE (no neighbours: the boundary layers see no ghosts)
        z_lo, z_hi = (int(v) for v in args.pmc_slab.split(","))
        c = capi.Context(n + 4096, params=p, device=args.pmc_device, slab=(z_lo, z_hi), ghost_capacity=1024)
    else:
        cfg = ic.CONFIGS[args.workload]
        c = capi.Context(n, box=cfg["box"], grid=cfg["grid"], device=args.pmc_device)
    with c:
        c.set_precision(args.precision == "mixed")
        c.load_snapshot(args.pmc_child)
        c.step(float(ic.DEFAULT_DT), args.warmup + args.steps)
        c.sync()


def pmc_traffic(ctx, args, steps=20, warmup=5, timeout_s=240, slab=None, device=0):
    """FETCH_SIZE and WRITE_SIZE of k_force<1,1,1> and k_density per launch, measured on the state of this run's timed window:
    the context is saved to a snapshot, and a child process (this script, --pmc-child) steps it under `rocprofv3 --pmc`,
    one pass per counter as MI355X_MICROARCH.md prescribes for gfx950 (the two cannot share a pass; --pmc alone, no trace
    domain).  Reads = FETCH_SIZE x 2 (wide coalesced reads are tallied at half their bytes), both in KB.  Returns
    ({kernel: bytes per launch}, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    work = tempfile.mkdtemp(prefix="sph_pmc_", dir="/tmp")
    snap = os.path.join(work, "flow.snap")
    out = {}
    try:
        ctx.save(snap)
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
            d = os.path.join(work, counter)
            names = [counter] if counter != "SQ" else ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"]
            cmd = [exe, "--pmc"] + names + ["--kernel-include-regex", "k_force|k_density", "--output-format", "csv", "-d", d, "-o", "p",
                   "--", sys.executable, os.path.abspath(__file__), "--pmc-child", snap, "--workload", args.workload, "--precision",
                   args.precision, "--steps", str(steps), "--warmup", str(warmup), "--pmc-device", str(device)]
            if slab is not None:
                cmd += ["--pmc-slab", f"{slab[0]},{slab[1]}"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr[-300:]}"
            per, name = {}, {}
            for row in csv.DictReader(open(files[0], newline="")):
                if row["Counter_Name"] not in names:
                    continue
                i = int(row["Dispatch_Id"])
                per.setdefault(i, {})
                per[i][row["Counter_Name"]] = per[i].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                name[i] = row["Kernel_Name"]
            for key, pat in (("force_fused", "k_force<"), ("density", "k_density")):       # (sph_step runs only the fused k_force<1,1,1>)
                ids = sorted(i for i in per if pat in name[i])[-steps:]
                if not ids:
                    return None, f"no {pat} dispatch in the {counter} pass"
                for c in names:
                    out.setdefault(key, {})[c] = sum(per[i].get(c, 0.0) for i in ids) / len(ids)
        res = {k: int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024) for k, v in out.items()}
        res["sq"] = {k: {c: v[c] for c in v if c.startswith("SQ_")} for k, v in out.items()}
        return res, (f"measured in THIS run: the state of the timed window saved to a snapshot and stepped {warmup} + {steps} times by a "
                     "child of this script under `rocprofv3 --pmc`, one pass each for FETCH_SIZE, WRITE_SIZE and the SQ_INSTS_* group (KB; reads = FETCH_SIZE x 2 "
                     f"per the gfx950 correction), means over the last {steps} launches; fabric requests of the L2s, Infinity-Cache hits "
                     "included: an upper bound on HBM traffic")
    except Exception as e:      # noqa: BLE001 -- the bench line must not depend on the profiler
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(work, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------
# single GPU
# ---------------------------------------------------------------------------------------------------------------
def _timed(ctx, dt, steps):
    ctx.sync()
    t0 = time.perf_counter()
    ctx.step(dt, steps)
    ctx.sync()
    return time.perf_counter() - t0


def _phases(ctx, dt, steps):
    ctx.timing(True); ctx.timing_reset()
    ctx.step(dt, steps)
    ctx.sync()
    ph, nst = ctx.timing_get()
    ctx.timing(False)
    return {k: v / max(nst, 1) for k, v in ph.items()}


def run_single(args):
    cfg = ic.CONFIGS[args.workload]
    n = cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2]
    dt = float(ic.DEFAULT_DT)
    ctx = capi.Context(n, box=cfg["box"], grid=cfg["grid"], device=0)
    ctx.set_precision(args.precision == "mixed")
    res = {"n": n, "cfg": cfg}

    # ---- state preparation: the dam after R steps (on the device, not timed) -----------------------------------
    ctx.reset_lattice(cfg["lattice"], jitter=True)        # synthetic data generated in HBM (== ic.dam_break_lattice)
    t0 = time.perf_counter()
    left = args.runup
    tail = min(1000, args.runup)                          # the last steps of the run-up characterise the state
    while left > 0:                                       # in pieces: a progress line per piece on stderr
        k = min(left - tail, 2000) if left > tail else left
        if left == tail:
            q0 = ctx.sort_stats()                         # (synchronises the stream)
            t_tail = time.perf_counter()
        ctx.step(dt, k); ctx.sync()
        left -= k
        print(f"[bench] run-up {args.runup - left}/{args.runup} steps, {time.perf_counter() - t0:.1f} s", file=sys.stderr,
              flush=True)
    res["runup_s"] = time.perf_counter() - t0
    if args.runup:
        tail_wall = time.perf_counter() - t_tail          # the last `tail` run-up steps, between two device syncs
        q1 = ctx.sort_stats()
        res["runup_tail"] = {"steps": tail, "movers_per_step": (q1["movers_total"] - q0["movers_total"]) / max(tail, 1),
                             "skips": q1["skips"] - q0["skips"], "wall_s": tail_wall,
                             "ms_per_step": tail_wall / max(tail, 1) * 1e3}
    else:
        res["runup_tail"] = {"steps": 0, "movers_per_step": 0.0, "skips": 0, "wall_s": 0.0, "ms_per_step": None}

    # ---- the headline: W warm-up + K timed steps of the flowing dam ------------------------------------------------
    ctx.step(dt, args.warmup)
    s0 = ctx.sort_stats()
    res["wall"] = _timed(ctx, dt, args.steps)
    s1 = ctx.sort_stats()
    res["sort"] = {"sorts": s1["sorts"] - s0["sorts"], "merges": s1["merges"] - s0["merges"],
                   "skips": s1["skips"] - s0["skips"],
                   "movers_per_step": (s1["movers_total"] - s0["movers_total"]) / max(args.steps, 1),
                   "last_movers": s1["last_movers"]}
    # per-phase device times (HIP events on the library's stream) from a second, instrumented pass over the same regime
    res["phases_ms"] = _phases(ctx, dt, args.steps)
    st = ctx.download(want=("density", "vel") if args.no_cpu else ("density", "vel", "pos"))
    if not args.no_pmc:                     # counter passes on exactly this state (a child process under rocprofv3)
        t_p = time.perf_counter()
        res["pmc"], res["pmc_note"] = pmc_traffic(ctx, args)
        print(f"[bench] PMC traffic passes: {time.perf_counter() - t_p:.1f} s ({'ok' if res['pmc'] else res['pmc_note']})",
              file=sys.stderr, flush=True)
    res["finite"] = bool(np.isfinite(st["density"]).all() and np.isfinite(st["vel"]).all())
    res["vmax"] = float(np.abs(st["vel"]).max())
    if not args.no_cpu:                     # the CPU legs step the SAME particles (by creation index) once
        res["flow_state"] = (np.ascontiguousarray(st["pos"][:, :3], dtype=np.float32),
                             np.ascontiguousarray(st["vel"][:, :3], dtype=np.float32))
    del st

    # ---- the same flowing state with the full radix sort every step --------------------------------------------------
    ctx.set_sort_mode(merge=False)
    ctx.step(dt, min(args.warmup, 3))
    res["wall_full_sort"] = _timed(ctx, dt, args.steps)
    res["phases_ms_full_sort"] = _phases(ctx, dt, min(args.steps, 20))
    ctx.set_sort_mode(merge=True)

    # ---- the dam at rest (first steps after the reset: nothing changes cell) ----------------------------------------------
    ctx.reset_lattice(cfg["lattice"], jitter=True)
    ctx.step(dt, args.warmup)
    r0 = ctx.sort_stats()
    res["wall_rest"] = _timed(ctx, dt, args.steps)
    r1 = ctx.sort_stats()
    res["sort_rest"] = {"sorts": r1["sorts"] - r0["sorts"], "skips": r1["skips"] - r0["skips"],
                        "movers_per_step": (r1["movers_total"] - r0["movers_total"]) / max(args.steps, 1)}
    res["phases_ms_rest"] = _phases(ctx, dt, min(args.steps, 20))
    ctx.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)       # >= 20 warm-up, >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--runup", type=int, default=None,
                    help="steps of state preparation before the warm-up (default: 6000 for C3 -- a flowing dam)")
    ap.add_argument("--workload", default="C3", choices=sorted(ic.CONFIGS))
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="--gpus N > 1: strong (default) = the SAME particles on every N -- BASELINE's metric is 'dam-break 16M "
                         "particles, 1/2/4/8 MI355X', i.e. --workload C3 (--workload C4 = config 4, 67,108,864 particles); "
                         "weak = 16.7 M particles PER GPU")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host", "local"],
                    help="--gpus N > 1: rccl = the library's own RCCL communicator (one rank per GPU); host = host-staged "
                         "messages (processes over gloo, or threads); local = device-to-device copies between the streams of "
                         "one process (--one-gpu without a launcher)")
    ap.add_argument("--lattice", default=None, help="slab path: nx,ny,nz instead of the workload's lattice (e.g. one rank's "
                                                    "eighth of C3: --force-slab --lattice 256,256,32)")
    ap.add_argument("--rebalance-every", type=int, default=500, help="slab path: re-balancing check every so many run-up steps")
    ap.add_argument("--precision", default="f32", choices=["f32", "mixed"],
                    help="mixed = BASELINE config 5's arithmetic (fp32 state, packed-fp16 pair arithmetic and per-row sums in "
                         "the density pass): a separate dtype line, never the fp32 headline")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes that measure roofline.traffic for this run")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-slab", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=24.0, help="seconds of host time for the cpu_baseline probe (four legs)")
    ap.add_argument("--protocol", type=int, default=3, choices=[1, 3],
                    help="slab path: 3 = MIGRANTS / HALO A / HALO B per step; 1 = the one-message step (two ghost layers, ghost "
                         "densities recomputed locally: sph_slab_set_protocol)")
    ap.add_argument("--force-slab", action="store_true", help="run the z-slab path even with one rank (testing)")
    ap.add_argument("--periodic-z", action="store_true",
                    help="with --force-slab: ONE slab between its own periodic images (loop transport) -- a middle rank's whole step, "
                         "halo work included, on one GPU; default lattice 256,256,32 (an eighth of C3)")
    ap.add_argument("--early-force", default="auto", choices=["auto", "on", "off"],
                    help="slab path (--gpus N, --periodic-z): sph_slab_set_early_force; auto = the rule a real run applies to its pings (on unless a group takes < 12 us and a halo-A message < 45 us)")
    ap.add_argument("--link-gbs", type=float, default=153.0, help="--periodic-z: bandwidth a message is held back for (0: no hold)")
    ap.add_argument("--link-latency-us", type=float, default=10.0, help="--periodic-z: latency a message is held back for")
    ap.add_argument("--one-gpu", action="store_true",
                    help="rehearsal of the multi-rank path on a one-GPU box: every rank uses device 0.  Without a launcher the "
                         "N ranks are N THREADS of this process (--transport local or host; a GPU box admits at most 6 "
                         "processes on its card); under torch.distributed.run they are processes (--transport host)")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    rank, world, local = _dist_env()
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ and not args.one_gpu:
        # called without a launcher: start one rank per GPU as child processes (nothing has touched the GPU yet)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if args.gpus > 1 or world > 1 or args.force_slab:
        from gpufluidsimulator_amd import slab
        return slab.bench_main(args)
    strict_flow = args.runup is None                       # the default configuration must be a flowing state
    if args.runup is None:
        args.runup = DEFAULT_RUNUP[args.workload]

    r = run_single(args)
    n, cfg, phases_ms = r["n"], r["cfg"], r["phases_ms"]
    value = n * args.steps / r["wall"]
    t_force, t_dens, t_sort = phases_ms["force"] * 1e-3, phases_ms["dens"] * 1e-3, phases_ms["sort"] * 1e-3
    achieved = BYTES_PER_PARTICLE["force_fused"] * n / t_force / 1e9
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure is the one
    # profiles/collect_pmc.sh measured for the same workload and state (separate rocprofv3 --pmc passes), and
    # `traffic_source` says so -- it is NOT a measurement of this run
    traffic, traffic_source, traffic_density = None, None, None
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if r.get("pmc"):
        traffic, traffic_density, traffic_source = r["pmc"]["force_fused"], r["pmc"].get("density"), r["pmc_note"]
    elif os.path.exists(prof):
        try:
            pj = json.load(open(prof))
            if pj.get("workload") == args.workload and pj.get("state") == "flow":
                traffic = pj.get("force_fused_hbm_bytes_per_launch")
                traffic_density = pj.get("density_hbm_bytes_per_launch")
                traffic_source = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  f"profiles/collect_pmc.sh ({pj.get('collected', 'date not recorded')}), FETCH_SIZE x 2 + "
                                  "WRITE_SIZE per the gfx950 correction; a constant of an earlier run of this command, not "
                                  f"of this process (the live passes were not taken: {r.get('pmc_note', '--no-pmc')})")
        except Exception:
            traffic = None
    # What binds the dominant kernel is VALU issue, not HBM: the instruction counts per wave are SQ
    # counters -- of this run's own rocprofv3 --pmc child when it ran (pmc_traffic), else of an earlier run of this command
    # (profiles/*_sq_counters.json) -- and the issue fraction prices them against THIS run's launch time
    issue = None
    sq = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_c3_flow_sq_counters.json"))
    live_sq = (r.get("pmc") or {}).get("sq", {}).get("force_fused")
    if live_sq or (sq and args.workload == "C3" and args.precision == "f32"):
        try:
            kj = live_sq or next(v for k, v in json.load(open(os.path.join(ROOT, "profiles", sq[-1]))).items() if k.startswith("k_force<true, true, true"))
            waves, clk_ghz, simds = kj["SQ_WAVES"], 2.4, 1024
            issue_s = kj["SQ_INSTS_VALU"] * 2.0 / simds / (clk_ghz * 1e9)      # one fp32 wave-instruction = 2 cycles of a SIMD
            issue = {"valu_insts_per_wave": kj["SQ_INSTS_VALU"] / waves, "lds_insts_per_wave": kj["SQ_INSTS_LDS"] / waves,
                     "salu_insts_per_wave": kj["SQ_INSTS_SALU"] / waves, "waves": int(waves),
                     "issue_ms_at_spec_clock": issue_s * 1e3, "issue_frac": issue_s / t_force,
                     "assumes": f"{simds} SIMDs, 2 cycles per fp32 wave-instruction, {clk_ghz} GHz spec clock (the chip holds ~2.0-2.1 "
                                "GHz at its power cap: the fraction at the held clock is ~1.17x this)",
                     "source": ("measured in THIS run: the SQ_INSTS_* pass of the rocprofv3 --pmc child (see traffic_source)" if live_sq else
                                f"profiles/{sq[-1]}: rocprofv3 --pmc SQ_INSTS_* of profiles/collect_pmc.sh over the timed window of "
                                "this command; a constant of an earlier run, not of this process")}
        except Exception:
            issue = None
    # a flowing state: no sort of the timed window was skipped, particles changed cell in it, and over the last 1000
    # run-up steps at least 1e-3 N of them did so per step (the dam falls as a lattice, so cell changes come in bursts:
    # a 20-step window can sit between two of them)
    flow_ok = (r["sort"]["skips"] == 0 and r["sort"]["movers_per_step"] >= 1e-4 * n
               and r["runup_tail"]["movers_per_step"] >= 1e-3 * n)
    out = {
        "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["wall"] / args.steps * 1e3,
        # the N = 1 point of the strong-scaling series BASELINE's metric names (the same 16.7 M particles on 1/2/4/8 GPUs)
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if args.precision == "f32" else "f32 state + packed-f16 density pairs (config 5's arithmetic: an option, slower than fp32 on gfx950)", "data": "synthetic",
        "config": {"workload": f"dam-break {args.workload}: {cfg['lattice'][0]}x{cfg['lattice'][1]}x{cfg['lattice'][2]} "
                               f"= {n} particles, grid {cfg['grid'][0]}^3, box {cfg['box'][0]}, dt 5e-7, jittered lattice"
                               f"{' (BASELINE config 4 on one GPU: the N = 1 point of --scaling strong)' if args.workload == 'C4' else ''}, "
                               f"FLOWING: timed after {args.runup} run-up steps on the device (+{args.warmup} warm-up)",
                   "particles": n, "grid": list(cfg["grid"]), "parallelism": "1 GPU, whole domain",
                   "state": "flow", "runup_steps": args.runup, "runup_seconds": r["runup_s"],
                   "runup_last_1000_steps": r["runup_tail"]},
        # `achieved` / `peak` / `frac` are the contract's HBM figure (algorithmic bytes over the launch time); `bound` names
        # what the evidence says binds the kernel: fp32 VALU instruction issue (`issue`, `valu`), not bandwidth
        "roofline": {"bound": "valu-issue", "kernel": "k_force<force+collision+integrate>", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_source, "issue": issue,
                     "algorithmic_bytes_per_particle": BYTES_PER_PARTICLE["force_fused"],
                     "avg_launch_ms": phases_ms["force"],
                     "valu": {"flop_per_particle": FLOP_PER_PARTICLE["force_fused"],
                              "achieved_tflops": FLOP_PER_PARTICLE["force_fused"] * n / t_force / 1e12,
                              "peak_tflops": VALU_PEAK_TFLOPS,
                              "frac": FLOP_PER_PARTICLE["force_fused"] * n / t_force / 1e12 / VALU_PEAK_TFLOPS},
                     "traffic_over_algorithmic": (traffic / (BYTES_PER_PARTICLE["force_fused"] * n)) if traffic else None,
                     "density_kernel": {"achieved": BYTES_PER_PARTICLE["dens"] * n / t_dens / 1e9, "unit": "GB/s",
                                        "traffic": traffic_density,
                                        "frac": BYTES_PER_PARTICLE["dens"] * n / t_dens / 1e9 / HBM_PEAK_GBS,
                                        "avg_launch_ms": phases_ms["dens"],
                                        "valu_tflops": FLOP_PER_PARTICLE["dens"] * n / t_dens / 1e12},
                     "sort_phase": {"achieved": BYTES_PER_PARTICLE["sort_merge"] * n / max(t_sort, 1e-9) / 1e9,
                                    "unit": "GB/s", "avg_ms": phases_ms["sort"],
                                    "frac": BYTES_PER_PARTICLE["sort_merge"] * n / max(t_sort, 1e-9) / 1e9 / HBM_PEAK_GBS}},
        # the sustained figure of the regime: the last 1000 run-up steps between two device syncs (cell changes come
        # in bursts -- whole lattice layers cross a face together -- and a 100-step window can sit in a lull)
        "value_sustained": (n * r["runup_tail"]["steps"] / r["runup_tail"]["wall_s"]) if r["runup_tail"]["wall_s"] else None,
        "ms_per_step_sustained": r["runup_tail"]["ms_per_step"],
        "phases_ms": phases_ms, "sort": r["sort"], "flowing": flow_ok, "finite": r["finite"], "vmax": r["vmax"],
        "value_full_sort": n * args.steps / r["wall_full_sort"],
        "ms_per_step_full_sort": r["wall_full_sort"] / args.steps * 1e3, "phases_ms_full_sort": r["phases_ms_full_sort"],
        "value_at_rest": n * args.steps / r["wall_rest"], "ms_per_step_at_rest": r["wall_rest"] / args.steps * 1e3,
        "phases_ms_at_rest": r["phases_ms_rest"], "sort_at_rest": r["sort_rest"],
    }
    if not args.no_cpu:
        pos_f, vel_f = r["flow_state"]
        label = (f"HEADLINE config: dam-break {args.workload}, {n} particles, grid {cfg['grid'][0]}^3, the flowing state of the "
                 f"GPU's timed window (after {args.runup} + {args.warmup} + 2 x {args.steps} steps), ONE step, one run; "
                 "team sizes from the 262,144-particle probe (small_sample)")
        cb = cpu_baseline(budget_s=args.cpu_budget, headline=(cfg, pos_f, vel_f, label))
        out["cpu_baseline"] = cb
        out["gpu_over_cpu"] = value / cb["value"]
    print(json.dumps(out), flush=True)
    if strict_flow and not flow_ok:
        sys.exit("bench: the timed window was not a flowing state (sort skipped or too few particles changed cell)")


if __name__ == "__main__":
    main()
