"""GPU: the JSON line the driver reads from `python bench.py` (N = 1), on a small workload so that it runs in a minute:
every key of the bench contract, the roofline block with its live counter passes (a rocprofv3 --pmc child on the run's own
state) and both cpu_baseline legs."""
import json
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_gpu_bench_line_keeps_the_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C2", "--runup", "400", "--steps", "6", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-1500:]                      # ONE JSON line on stdout
    d = json.loads(lines[0])
    assert d["metric"] == "particle-steps/sec" and d["unit"] == "particle-steps/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] in ("strong", "weak") and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "262144 particles" in d["config"]["workload"]
    assert abs(d["value"] - 262144 * 6 / (d["ms_per_step"] * 6e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "valu-issue" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.0 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - 84 * 262144 / (rf["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * rf["achieved"]
    if shutil.which("rocprofv3"):                                # the counter passes of THIS run
        assert "measured in THIS run" in rf["traffic_source"], rf["traffic_source"]
        assert rf["traffic"] >= 0.5 * 84 * 262144 and rf["traffic_over_algorithmic"] == rf["traffic"] / (84 * 262144)
        assert rf["issue"]["waves"] == 262144 // 64 and 4000 < rf["issue"]["valu_insts_per_wave"] < 12000
        assert "measured in THIS run" in rf["issue"]["source"] and 0.0 < rf["issue"]["issue_frac"] < 1.5
        assert rf["density_kernel"]["traffic"] >= 0.5 * 20 * 262144
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "particle-steps/s"
    assert "262144" in cb["sample"] or "C2" in cb["sample"]
    assert d["gpu_over_cpu"] == pytest.approx(d["value"] / cb["value"])
    assert d["finite"] is True and d["sort"]["skips"] >= 0
