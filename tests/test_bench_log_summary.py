"""Row f4: the benchmark-log summariser has the semantics of the reference's own parser
(/root/reference/benchmark.py:4-67: skip two header lines, mean of every field, each phase as % of total) and reads
BOTH line forms -- the one of the reference's committed logs (benchmarks/oscar/<N>/*.txt, 'FPS:..fps') and the one its
current source writes (SPH/particleSystem.cpp:697-716, 'frames:..frames'), which is also what this build writes.

The fixtures under tests/golden/oscar_logs/ are four of the reference's committed logs (data: the numbers BASELINE.md
section 1 quotes); the expected values below are BASELINE.md's, which were computed by hand from the same files.
"""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "bench_log_summary.py")
LOGS = os.path.join(ROOT, "tests", "golden", "oscar_logs")

spec = importlib.util.spec_from_file_location("bench_log_summary", TOOL)
bls = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bls)


def test_published_cuda_log_gives_the_figures_of_baseline_md():
    s = bls.summarise(os.path.join(LOGS, "n131072_CUDA.txt"))
    assert s["mode"] == "CUDA" and s["samples"] == 23 and s["style"] == "oscar"
    us = {k: v / 1e3 for k, v in s["mean_ns"].items()}
    # BASELINE.md section 1, row "CUDA step, N=131 072" and the phase row below it
    assert us["total"] == pytest.approx(3543.3, abs=0.05)
    assert us["sort"] == pytest.approx(3181.0, abs=0.05)
    assert us["b'-grid"] == pytest.approx(247.2, abs=0.05)
    assert us["copying"] == pytest.approx(28.2, abs=0.05) and us["z-index"] == pytest.approx(9.3, abs=0.05)
    assert us["b-grid"] == pytest.approx(9.0, abs=0.05) and us["integrate"] == pytest.approx(2.1, abs=0.05)
    assert us["dens"] == pytest.approx(1.9, abs=0.05) and us["force"] == pytest.approx(1.6, abs=0.05)
    assert us["collision"] == pytest.approx(1.5, abs=0.05)
    # % of total, as benchmark.py:58-67 prints them for this file: sort 89.78, b'-grid 6.98, copying 0.8
    p = s["percent_of_total"]
    assert round(p["sort"], 2) == 89.78 and round(p["b'-grid"], 2) == 6.98 and round(p["copying"], 2) == 0.8
    assert "total" not in p and "fps" not in p


def test_published_omp_and_small_cuda_logs():
    s = bls.summarise(os.path.join(LOGS, "n131072_OMP.txt"))
    ms = {k: v / 1e6 for k, v in s["mean_ns"].items()}
    # OMP step 255.4 ms; force 110.4, dens 82.5, collision 36.4, sort 20.6, z-index 3.92
    assert ms["total"] == pytest.approx(255.4, abs=0.06) and ms["force"] == pytest.approx(110.4, abs=0.05)
    assert ms["dens"] == pytest.approx(82.5, abs=0.05) and ms["collision"] == pytest.approx(36.4, abs=0.05)
    assert ms["sort"] == pytest.approx(20.6, abs=0.05) and ms["z-index"] == pytest.approx(3.92, abs=0.005)
    assert 131072 / (s["mean_ns"]["total"] * 1e-9) == pytest.approx(0.513e6, rel=2e-3)      # particle-steps/s
    s = bls.summarise(os.path.join(LOGS, "n8192_CUDA.txt"))
    us = {k: v / 1e3 for k, v in s["mean_ns"].items()}
    assert us["total"] == pytest.approx(1298.9, abs=0.05) and us["b'-grid"] == pytest.approx(921.9, abs=0.05)
    assert us["sort"] == pytest.approx(294.5, abs=0.05)


def test_sequential_log_without_grid_fields_and_without_a_decimal_point_in_fps():
    """'FPS:0fps' is what the reference's sequential logs hold; its own regex needs 'd.d' there and finds no line."""
    s = bls.summarise(os.path.join(LOGS, "n131072_sequential.txt"))
    assert s["samples"] == 1 and s["mode"] == "sequential"
    assert s["mean_ns"]["total"] / 1e9 == pytest.approx(506.4, abs=0.05)                    # 506.4 s
    assert "sort" not in s["mean_ns"] and set(s["percent_of_total"]) == {"dens", "force", "collision", "integrate"}
    assert 131072 / (s["mean_ns"]["total"] * 1e-9) == pytest.approx(259, abs=1)


def test_both_line_forms_of_this_build(tmp_path):
    """The two forms ParticleSystem::setBenchmarkLog writes (csrc/particleSystem.cpp), made up here: same numbers,
    different head and tail -- same summary."""
    body = ("\ttotal:{t}ns,\t\tcopying:0ns,\t\tz-index:0ns,\t\tsort:{s}ns,\t\tb-grid:0ns,\t\tb'-grid:0ns,\t\tdens:{d}ns,"
            "\t\tforce:{f}ns,\t\tcollision:0ns,\t\tintegrate:0ns,\t\t")
    rows = [(100000, 20000, 30000, 50000), (120000, 30000, 30000, 60000), (80000, 10000, 30000, 40000)]
    frames = tmp_path / "frames.txt"
    oscar = tmp_path / "oscar.txt"
    with open(frames, "w") as a, open(oscar, "w") as b:
        for f in (a, b):
            f.write("SPH Particle Simulation Benchmark\nCompute mode: HIP\n")
        for k, (t, s, d, fo) in enumerate(rows):
            line = body.format(t=t, s=s, d=d, f=fo)
            a.write(f"{2 * (k + 1)}sec" + line + f"frames:{40 * (k + 1)}frames\n")
            b.write(f"{2.0 * (k + 1) + 0.004:.3f}sec" + line + f"FPS:{9000.0 + k:.3f}fps\n")
    sa, sb = bls.summarise(str(frames)), bls.summarise(str(oscar))
    assert sa["style"] == "frames" and sb["style"] == "oscar" and sa["mode"] == sb["mode"] == "HIP"
    for k in ("total", "sort", "dens", "force"):
        assert sa["mean_ns"][k] == sb["mean_ns"][k]
    assert sa["mean_ns"]["total"] == 100000 and sa["percent_of_total"]["force"] == 50.0
    assert sa["mean_ns"]["frames"] == 80 and sb["mean_ns"]["fps"] == pytest.approx(9001.0)
    # the command line: a table for people, --json for tools
    out = subprocess.run([sys.executable, TOOL, "--json", str(frames), str(oscar)], capture_output=True, text=True, check=True)
    got = json.loads(out.stdout)
    assert [g["style"] for g in got] == ["frames", "oscar"] and got[0]["mean_ns"]["total"] == 100000
    txt = subprocess.run([sys.executable, TOOL, str(oscar)], capture_output=True, text=True, check=True).stdout
    assert "total" in txt and "us per update" in txt and "50.00 %" in txt


def test_not_a_log(tmp_path):
    p = tmp_path / "x.txt"
    p.write_text("SPH Particle Simulation Benchmark\nCompute mode: HIP\nnothing here\n")
    with pytest.raises(ValueError):
        bls.summarise(str(p))
