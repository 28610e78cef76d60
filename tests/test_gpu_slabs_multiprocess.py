"""GPU: the slab path as SEPARATE PROCESSES (torch.distributed.run, one rank per slab) with the product
engine -- the launch shape of `bench.py --gpus N`.  The test box has one GPU, so both ranks use cuda:0 and
the process group is gloo (tensors staged through the host); on an N-GPU node the same driver runs over
RCCL with device tensors.  The result must match the whole-domain context."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- before libsph_hip.so is loaded (capi.load)

from gpufluidsimulator_amd import capi
from slab_oracle_engine import make_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("case,world", [("up", 2), ("shear", 3)])
def test_slab_processes_match_whole_domain(case, world, tmp_path):
    steps = 24
    out = str(tmp_path / "slabs.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "slab_gpu_worker.py"),
           "--case", case, "--steps", str(steps), "--out", out]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(out)
    pos, vel, box, grid = make_case(case)
    with capi.Context(pos.shape[0], box=box, grid=grid) as c:
        c.upload(pos, vel)
        c.step(5e-7, steps)
        ref = c.download()
    migrants, _, owned = (int(v) for v in got["stats"])
    assert owned == pos.shape[0]
    if case == "up":
        assert migrants > 0                     # particles crossed the cut and changed process
    assert np.abs(got["pos"] - ref["pos"]).max() <= 1e-6 * max(box)
    assert np.abs(got["vel"] - ref["vel"]).max() <= 1e-5 * np.abs(ref["vel"]).max()
    assert np.abs(got["density"] / ref["density"] - 1).max() <= 1e-5
