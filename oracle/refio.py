"""File protocol of oracle/_ref/sph_ref (see oracle/ref_harness.cpp) -- TEST INFRASTRUCTURE."""
from __future__ import annotations

import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(HERE, "_ref", "sph_ref")
DROPIN_BIN = os.path.join(HERE, "_ref", "sph_ref_dropin")   # reference host code + libsph_hip seam (GPU)

TAGS = {1: "zindex", 2: "order", 3: "sorted_z", 4: "bcells", 5: "bprime", 6: "dens", 7: "force", 8: "coll",
        9: "state", 10: "hpos"}
_FLOAT_TAGS = {6, 7, 8, 9, 10}


def available() -> bool:
    return os.path.exists(REF_BIN)


def dropin_available() -> bool:
    return os.path.exists(DROPIN_BIN)


def run_ref(pos, vel, box, grid, dt, steps, phases=False, dump_steps=(), threads=0, timeout=3600, binary=None):
    """Run the reference's OMP-mode step ``steps`` times.  Returns (records, stats):
    records[(name, step)] -> ndarray, stats = the JSON line the binary prints."""
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    vel = np.ascontiguousarray(vel, dtype=np.float32).reshape(-1, 3)
    n = pos.shape[0]
    if np.isscalar(box):
        box = (box,) * 3
    with tempfile.TemporaryDirectory(prefix="sphref_") as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            np.array([0x49485053, n], dtype=np.uint32).tofile(f)
            np.array(box, dtype=np.float32).tofile(f)
            np.array([int(grid)], dtype=np.uint32).tofile(f)
            np.array([dt], dtype=np.float32).tofile(f)
            np.array([steps, 1 if phases else 0], dtype=np.uint32).tofile(f)
            pos.tofile(f)
            vel.tofile(f)
        cmd = [binary or REF_BIN, fin, fout]
        if dump_steps:
            cmd.append("dump_steps=" + ",".join(str(int(s)) for s in dump_steps))
        if threads:
            cmd.append(f"threads={int(threads)}")
        out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=timeout)
        stats = json.loads(out.stdout.strip().splitlines()[-1])
        raw = np.fromfile(fout, dtype=np.uint32)
    assert raw[0] == 0x4F485053 and raw[1] == n
    nrec = int(raw[2])
    recs, off = {}, 3
    for _ in range(nrec):
        tag, step, count, width = (int(v) for v in raw[off:off + 4])
        off += 4
        body = raw[off:off + count * width]
        off += count * width
        arr = body.view(np.float32) if tag in _FLOAT_TAGS else body
        recs[(TAGS[tag], step)] = arr.reshape(count, width).copy() if width > 1 else arr.copy()
    return recs, stats
