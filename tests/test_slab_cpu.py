"""CPU: the multi-GPU z-slab protocol (gpufluidsimulator_amd.slab) with CPU ranks.

The product engine is the HIP library; here a test-only engine with the oracle behind it
(tests/slab_oracle_engine.py) is injected so that partitioning, count messages, migration, ghost
layers and the density halo are exercised with world_size > 1 on CPU -- once with in-process ranks
(LocalComm) and once with real processes over torch.distributed's gloo backend."""
import os
import tempfile
import threading

import numpy as np
import pytest

from gpufluidsimulator_amd import slab
from oracle import oracle
from slab_oracle_engine import OracleEngine, OracleEngine2, gloo_worker, make_case

DT = 5e-7


def _single_domain(case, steps):
    pos, vel, box, grid = make_case(case)
    o = oracle.Oracle(pos, vel, box, grid, oracle.CELL_LINEAR)
    o.step(DT, steps)
    st = o.state()
    o.close()
    return st, box


def _check(st, ref, box):
    assert np.abs(st["pos"] - ref["pos"]).max() <= 1e-6 * max(box)
    assert np.abs(st["vel"] - ref["vel"]).max() <= 1e-5 * np.abs(ref["vel"]).max()
    assert np.abs(st["density"] / ref["density"] - 1).max() <= 1e-5


def test_choose_cuts_balances_counts():
    hist = np.zeros(64, dtype=np.int64)
    hist[0:16] = 1000                       # dam break: a quarter of the layers hold everything
    cuts = slab.choose_cuts(hist, 4)
    assert cuts == [0, 4, 8, 12, 64]
    cuts = slab.choose_cuts(hist, 8)
    assert cuts[0] == 0 and cuts[-1] == 64 and all(b > a for a, b in zip(cuts, cuts[1:]))
    counts = [hist[a:b].sum() for a, b in zip(cuts, cuts[1:])]
    assert max(counts) == min(counts) == 2000
    with pytest.raises(ValueError):
        slab.choose_cuts(np.ones(3), 4)
    # degenerate: everything in one layer -> every slab still gets >= 2 layers (the halo protocol needs its two
    # boundary layers to be different layers)
    h = np.zeros(8, dtype=np.int64); h[5] = 10
    cuts = slab.choose_cuts(h, 3)
    assert all(b - a >= 2 for a, b in zip(cuts, cuts[1:])) and cuts[-1] == 8
    with pytest.raises(ValueError):
        slab.choose_cuts(np.ones(7), 4)     # 4 slabs x 2 layers do not fit 7 layers
    assert slab.choose_cuts(np.ones(8), 4) == [0, 2, 4, 6, 8]


def test_cell_layer_matches_the_oracle_hash():
    pos, vel, box, grid = make_case("up")
    o = oracle.Oracle(pos, vel, box, grid, oracle.CELL_LINEAR)
    o.map_zindex()
    z = o.particles["zindex"] // (grid[0] * grid[1])
    assert np.array_equal(slab.cell_layer_of(pos[:, 2], box[2], grid[2]), z)
    o.close()


@pytest.mark.parametrize("case,world", [("up", 3), ("down", 2), ("shear", 4)])
def test_slabs_in_process_match_single_domain(case, world):
    steps = 24
    pos, vel, box, grid = make_case(case)
    hub = slab.LocalComm.Hub(world)
    results, errors = [None] * world, []

    def rank_main(r):
        try:
            sim = slab.SlabSimulation(slab.LocalComm(hub, r), OracleEngine, box, grid, particles=(pos, vel))
            sim.run(DT, steps)
            results[r] = (sim.gather_state(), dict(sim.stats), sim.cuts)
        except BaseException as e:     # noqa: BLE001 - surface the failure and release the other ranks
            errors.append(e)
            hub.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=600)
    assert not errors, errors
    st, stats, cuts = results[0]
    ref, _ = _single_domain(case, steps)
    _check(st, ref, box)
    total_migrants = sum(res[1]["migrants"] for res in results)
    assert total_migrants > 0, "the case is meant to push particles across slab boundaries"
    for other in results[1:]:
        assert np.array_equal(other[0]["pos"], st["pos"])


@pytest.mark.parametrize("protocol", [3, 1])
def test_slabs_over_gloo_world_size_2(protocol):
    """Two real processes over torch.distributed's gloo backend -- under the three-group protocol and under the one-message
    protocol (slab.SlabSimulation._step_one, the Python statement of csrc/sph_slab.hip's protocol 1: one fixed-size message
    per neighbour and step, ghost densities recomputed by the receiver)."""
    import torch.multiprocessing as mp
    steps = 16
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000) + protocol
        mp.spawn(gloo_worker, args=(2, port, "tall_up" if protocol == 1 else "up", steps, d, protocol), nprocs=2, join=True)
        out = np.load(os.path.join(d, "out.npz"))
        stats = np.load(os.path.join(d, "stats.npy"))
    ref, box = _single_domain("tall_up" if protocol == 1 else "up", steps)
    _check(out, ref, box)
    assert stats[0] > 0
    assert stats[2] == (2 * (steps - 1) if protocol == 1 else 0)                 # one-message steps, summed over the two ranks


@pytest.mark.parametrize("case,world", [("up", 3), ("shear", 3), ("down", 2)])
def test_one_message_protocol_in_process_matches_single_domain(case, world):
    """The one-message protocol with in-process CPU ranks and the oracle engine (two ghost layers): the first step speaks the
    three-group protocol and learns the counts, every later step is ONE exchange (+ the rest of a message that outgrew its
    size); no density message at all -- the receiving rank's own density pass covers its ghosts."""
    steps = 24
    pos, vel, box, grid = make_case(case)
    hub = slab.LocalComm.Hub(world)
    results, errors = [None] * world, []

    def rank_main(r):
        try:
            sim = slab.SlabSimulation(slab.LocalComm(hub, r), OracleEngine2, box, grid, particles=(pos, vel), python_protocol=1)
            assert all(b - a >= 4 for a, b in zip(sim.cuts, sim.cuts[1:]))
            sim.run(DT, steps)
            results[r] = (sim.gather_state(), dict(sim.stats), sim.cuts)
        except BaseException as e:     # noqa: BLE001
            errors.append(e)
            hub.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=600)
    assert not errors, errors
    st, stats, cuts = results[0]
    ref, _ = _single_domain(case, steps)
    _check(st, ref, box)
    assert sum(res[1]["migrants"] for res in results) > 0
    assert all(res[1]["one_steps"] == steps - 1 for res in results), [res[1] for res in results]


def test_one_message_rule_and_cuts():
    """The size rule both ends of a link apply to the previous step's counts (csrc/sph_slab.hip: one_cap), and cuts of at
    least four layers for the protocol that sends two of them to either side."""
    f = slab.SlabSimulation.one_message_rows
    assert f(0) == 1024 and f(1000) == 2112 and f(262144) == 279552 and all(f(k) % 64 == 0 and f(k) >= k + 1024 for k in (1, 63, 4097, 10 ** 6))
    hist = np.zeros(64, dtype=np.int64); hist[0:16] = 1000
    assert slab.choose_cuts(hist, 4, 4) == [0, 4, 8, 12, 64]
    cuts = slab.choose_cuts(hist, 8, 4)
    assert all(b - a >= 4 for a, b in zip(cuts, cuts[1:])) and cuts[-1] == 64
    with pytest.raises(ValueError):
        slab.choose_cuts(np.ones(15), 4, 4)


def test_rebalance_moves_the_cuts_and_keeps_the_physics():
    """A column of fluid leaves its slabs (uniform upward drift): static cuts end up with one rank holding
    most particles; rebalance() re-cuts by count, ships whole layers point to point, and the run goes on
    matching the single-domain oracle."""
    case, world, steps = "up", 3, 65      # re-cuts after steps 20, 40, 60; densities are those of the last step
    pos, vel, box, grid = make_case(case)
    vel[:, 2] = 12000.0                         # 0.006 per step: about six cell layers over the run
    hub = slab.LocalComm.Hub(world)
    results, errors = [None] * world, []

    def rank_main(r):
        try:
            sim = slab.SlabSimulation(slab.LocalComm(hub, r), OracleEngine, box, grid, particles=(pos, vel))
            cuts0 = list(sim.cuts)
            sim.run(DT, steps, rebalance_every=20)
            results[r] = (sim.gather_state(), dict(sim.stats), cuts0, list(sim.cuts), sim.engine.n)
        except BaseException as e:     # noqa: BLE001
            errors.append(e)
            hub.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=900)
    assert not errors, errors
    st, stats, cuts0, cuts1, _ = results[0]
    assert stats.get("rebalances", 0) >= 1 and cuts1 != cuts0
    owned = [r[4] for r in results]
    assert sum(owned) == pos.shape[0] and max(owned) <= 1.35 * pos.shape[0] / world, owned
    o = oracle.Oracle(pos, vel, box, grid, oracle.CELL_LINEAR)
    o.step(DT, steps)
    ref = o.state(); o.close()
    _check(st, ref, box)


def test_bench_config_is_strong_scaling_of_config_3_by_default():
    """BASELINE.json's metric: "dam-break 16M particles, 1/2/4/8 MI355X" -- the SAME 16,777,216 particles on every N.
    `bench.py --gpus N` must default to that (round 2 defaulted to 16.7 M particles PER GPU); config 4 and weak scaling
    stay available; at 8 ranks the count-balanced cuts of the C3 lattice give every rank 16 cell layers."""
    import argparse
    from gpufluidsimulator_amd import ic
    a = argparse.Namespace(scaling="strong", workload="C3", lattice=None)
    for world in (1, 2, 4, 8):
        cfg, strong, label = slab.bench_config(a, world)
        assert strong and cfg["lattice"] == (256, 256, 256) and cfg["grid"] == (512, 512, 512)
        assert cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2] == 16777216
    cfg, strong, _ = slab.bench_config(argparse.Namespace(scaling="strong", workload="C4", lattice=None), 8)
    assert strong and cfg["lattice"][0] * cfg["lattice"][1] * cfg["lattice"][2] == 67108864
    cfg, strong, _ = slab.bench_config(argparse.Namespace(scaling="weak", workload="C3", lattice=None), 4)
    assert not strong and cfg["lattice"] == (256, 256, 1024)
    cfg, _, _ = slab.bench_config(argparse.Namespace(scaling="strong", workload="C3", lattice="256,256,32"), 1)
    assert cfg["lattice"] == (256, 256, 32)
    assert ic.CONFIGS["C5"]["lattice"] == (512, 512, 512) and ic.CONFIGS["C5"]["grid"] == (1024, 1024, 1024)
    # the lattice is cell-aligned (2 lattice layers per cell layer): 256 lattice layers fill cell layers 0..127
    hist = np.zeros(512, dtype=np.int64)
    hist[:128] = 2 * 256 * 256
    cuts = slab.choose_cuts(hist, 8)
    assert cuts == [0, 16, 32, 48, 64, 80, 96, 112, 512]
    assert all(b - a >= 7 for a, b in zip(cuts, cuts[1:]))        # every slab has a deep interior (>= 7 owned layers)


def test_single_hop_cuts_reach_any_target_one_rank_at_a_time():
    """slab.single_hop_cuts: the sequence of cuts a device-side re-cut (sph_slab_recut moves a particle at most one rank per
    call) goes through; every hop keeps new cut r within [old cut r-1, old cut r+1], the order and the minimum thickness."""
    rng = np.random.default_rng(3)
    for _ in range(300):
        world, gz = int(rng.integers(2, 9)), 64
        def cuts():
            inner = np.sort(rng.choice(np.arange(2, gz - 1, 2), world - 1, replace=False))
            return [0] + [int(v) for v in inner] + [gz]
        old, new = cuts(), cuts()
        hops = 0
        while old != new:
            step = slab.single_hop_cuts(old, new)
            assert step != old and step[0] == 0 and step[-1] == gz
            assert all(old[r - 1] <= step[r] <= old[r + 1] for r in range(1, world))
            assert all(b - a >= 2 for a, b in zip(step, step[1:])), (old, new, step)
            old, hops = step, hops + 1
            assert hops <= world + 2


def test_gather_rows_puts_every_rank_s_numbers_on_every_rank():
    """slab.gather_rows (what bench_diagnostics uses to print EVERY rank's phases): one all-reduce, NaN = 'none' survives."""
    import threading
    world = 4
    hub = slab.LocalComm.Hub(world)
    out = [None] * world

    def rank_main(r):
        comm = slab.LocalComm(hub, r)
        out[r] = slab.gather_rows(comm, [float(r), 10.0 * r, float("nan") if r % 2 else 1.5])

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts: t.start()
    for t in ts: t.join(timeout=60)
    for m in out:
        assert m.shape == (world, 3)
        assert np.array_equal(m[:, 0], np.arange(world)) and np.array_equal(m[:, 1], 10.0 * np.arange(world))
        assert np.isnan(m[1, 2]) and np.isnan(m[3, 2]) and m[0, 2] == 1.5 and m[2, 2] == 1.5


def test_early_force_rule_follows_the_measured_crossovers():
    """slab.NativeSlabSimulation.early_force_rule: the launchers' "auto" for sph_slab_set_early_force, from the preflight pings and
    the rank's particle count.  The cases are the measured ones (periodic-slab harness)."""
    rule = slab.NativeSlabSimulation.early_force_rule
    n8 = 2097152                                    # an eighth of config 3
    assert rule(16.0, 16.0, n8)[0] is False         # device copies only: +6 us per step with it
    assert rule(19.0, 47.0, n8)[0] is False         # 10 us + 153 GB/s: 7 us per step slower with it (round 6's last table)
    assert rule(28.0, 57.0, n8)[0] is False         # 20 us per group: still 5 us slower
    assert rule(48.0, 79.0, n8)[0] is True          # 40 us per group: -30 us per step
    assert rule(17.0, 74.0, n8)[0] is False         # 10 us + 75 GB/s: level
    assert rule(5.0, 130.0, n8)[0] is True          # a slow link rather than a late one
    assert rule(22.0, 147.0, 16777216)[0] is False  # a config-5 rank: its deep density launch outlasts the messages
    assert rule(91.0, 214.0, 16777216)[0] is False
    on, why = rule(19.0, 47.0, n8)
    assert "covers" in why and "19.0" in why
