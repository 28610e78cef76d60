"""One rank of the two-process GPU rehearsal (tests/test_gpu_slabs_multiprocess.py), started by
`python -m torch.distributed.run`: the product engine (HipEngine -> libsph_hip.so) on cuda:0, one
process per slab, torch.distributed point-to-point between the processes.  The box has one GPU, so the
backend is gloo with host staging (TorchDistComm.staged); everything else -- process group, slab
driver, pack/unpack kernels, migration, ghost layers, density halo -- is what `bench.py --gpus N` runs."""
import argparse
import os
import sys

import numpy as np
import torch                                    # before libsph_hip.so (capi.load)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpufluidsimulator_amd import slab  # noqa: E402
from slab_oracle_engine import make_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="up")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    dist.init_process_group("gloo")
    try:
        pos, vel, box, grid = make_case(a.case)
        comm = slab.TorchDistComm(torch.device("cuda", 0))
        # the product step (sph_slab_step); two ranks cannot share a GPU under RCCL, so the messages travel host-staged
        # over gloo -- on an N-GPU node the transport is "rccl"
        sim = slab.NativeSlabSimulation(comm, box, grid, device_index=0, transport="host", particles=(pos, vel))
        sim.run(5e-7, a.steps)
        st = sim.gather_state()
        assert sim.stats["host_waits"] == a.steps
        stats = comm.allreduce_sum(np.array([sim.stats["migrants"], sim.stats["resorts"], sim.engine.n], dtype=np.int64))
        if dist.get_rank() == 0:
            np.savez(a.out, cuts=np.array(sim.cuts), stats=stats, **st)
        sim.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
