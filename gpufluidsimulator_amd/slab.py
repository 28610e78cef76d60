"""z-slab domain decomposition of the SPH step across the GPUs of one node.

No counterpart in the reference (single GPU): this is new work on top of the
C ABI's halo entry points (include/sph_hip.h, csrc/sph_halo.hip).

One process per GPU.  Rank r owns the cell layers [cuts[r], cuts[r+1]) of the global grid (cuts are
count-balanced from a per-layer particle histogram, because a dam break fills only a quarter of
the box).  Interaction reach is one cell layer (3x3x3 stencil), so per step a rank exchanges with
its two z-neighbours only, point to point (torch.distributed send/recv = RCCL over the direct xGMI
link); no collective sits on the data path:

    hash + sort owned particles           (local keys: z slowest, so leavers sit at the array ends)
    [counts]    one tiny message per neighbour: {#migrants to you, #my boundary-layer particles}
    [migrants]  particles whose cell left the slab move to the neighbour (8-float records)
    [halo A]    boundary layers -> neighbour's ghost layers (positions, velocities, indices)
    cell table over ghosts + owned ; density pass over owned
    [halo B]    (density, pressure) of the same boundary particles
    fused force + collision + integrate over owned

That is the THREE-GROUP protocol (SlabSimulation.step; in the library: csrc/sph_slab.hip, the default).  The ONE-MESSAGE protocol
(SlabSimulation._step_one; in the library: sph_slab_set_protocol(s, 1), NativeSlabSimulation(protocol=1)) sends header, leavers and
two layers of residents in one message per neighbour whose size was fixed from the previous step's counts, and the receiver
recomputes the ghost densities itself: no density message.

The per-rank compute engine is pluggable ONLY so that tests can drive this protocol on CPU ranks
(gloo) with the oracle behind it; the product engine is `HipEngine` (libsph_hip.so) and nothing
else is ever chosen implicitly: without a gfx950 device `HipEngine` raises.
"""
from __future__ import annotations

import json
import os
import queue
import sys
import threading
import time

import numpy as np

from . import capi, ic

REC = capi.HALO_RECORD_FLOATS


# ------------------------------------------------------------------------------------------------
# partitioning
# ------------------------------------------------------------------------------------------------
def cell_layer_of(z, box_z, gz):
    """Global z cell layer with the device's arithmetic (csrc/sph_device.hpp cell_coord): subtract
    boxMin, divide by the box dimension, multiply by the grid size, floor, clamp -- all in fp32."""
    z = np.asarray(z, dtype=np.float32)
    bmin = np.float32(-np.float32(box_z) / np.float32(2.0))
    bdim = np.float32(np.float32(box_z) / np.float32(2.0)) - bmin
    q = ((z - bmin) / bdim) * np.float32(gz)
    return np.clip(np.floor(q).astype(np.int64), 0, int(gz) - 1)


MIN_SLAB_LAYERS = 2


def choose_cuts(hist, world, min_layers=None):
    """Count-balanced slab boundaries: cuts[r] .. cuts[r+1] are rank r's cell layers.
    `min_layers`: 2 (default, below) or 4 for the one-message slab step (sph_slab_set_protocol: the two layers sent to either
    neighbour must be four different layers).

    Every slab keeps at least MIN_SLAB_LAYERS = 2 layers: the halo protocol treats a slab's lowest and highest
    layer as two different boundary layers (what arrives from below lands in the first, what arrives from above in
    the second); with a one-layer slab they would be the same layer and a neighbour would be sent too few ghosts."""
    hist = np.asarray(hist, dtype=np.int64)
    gz, total = hist.shape[0], int(hist.sum())
    m = (min_layers or MIN_SLAB_LAYERS) if world > 1 else 1
    if world * m > gz:
        raise ValueError(f"{world} ranks need {m} cell layers each but the grid has only {gz}")
    prefix = np.concatenate([[0], np.cumsum(hist)])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        z = int(np.searchsorted(prefix, target, side="left"))
        # the boundary that splits the counts best: z or z-1
        if z > 0 and abs(prefix[z - 1] - target) <= abs(prefix[min(z, gz)] - target):
            z -= 1
        z = max(z, cuts[-1] + m)              # every slab keeps at least m layers ...
        z = min(z, gz - m * (world - r))      # ... and leaves as many to each rank above
        cuts.append(z)
    cuts.append(gz)
    assert all(b - a >= m for a, b in zip(cuts, cuts[1:])), cuts
    return [int(c) for c in cuts]


def single_hop_cuts(old, new):
    """The cuts one `sph_slab_recut` call may go to on the way from `old` to `new`: a re-cut moves every particle at most
    ONE rank, i.e. new cut r must lie within [old cut r-1, old cut r+1] (what rank r gives away below its new lower cut
    must belong to rank r-1's new layers, and the same above).  Clipping every target cut into that interval keeps the
    cuts ordered and the slabs at least MIN_SLAB_LAYERS thick (both `old` and `new` are); repeat until old == new."""
    old, new = [int(c) for c in old], [int(c) for c in new]
    assert len(old) == len(new) and old[0] == new[0] and old[-1] == new[-1]
    out = [old[0]]
    for r in range(1, len(old) - 1):
        out.append(min(max(new[r], old[r - 1]), old[r + 1]))
    out.append(old[-1])
    return out


# ------------------------------------------------------------------------------------------------
# communication back ends (same three calls)
# ------------------------------------------------------------------------------------------------
class TorchDistComm:
    """torch.distributed point-to-point; backend "nccl" is RCCL on ROCm (xGMI), "gloo" on CPU."""

    def __init__(self, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        # gloo moves host memory only: device tensors are staged through the host (rehearsals of the
        # multi-process path on a box with fewer GPUs than ranks; RCCL takes device tensors as they are)
        self.staged = dist.get_backend() == "gloo" and device.type != "cpu"
        self.wire = torch.device("cpu") if self.staged else device

    def exchange(self, sends, recvs):
        """sends/recvs: lists of (peer, tensor); returns when all of them have completed."""
        dist = self.dist
        recvs = [(p, t) for p, t in recvs if t.numel()]
        sends = [(p, t) for p, t in sends if t.numel()]
        if self.staged:
            landing = [(p, t, self.torch.empty(t.shape, dtype=t.dtype)) for p, t in recvs]
            ops = [dist.P2POp(dist.irecv, h, p) for p, _, h in landing]
            ops += [dist.P2POp(dist.isend, t.cpu(), p) for p, t in sends]     # .cpu() orders after the pack kernels
        else:
            ops = [dist.P2POp(dist.irecv, t, p) for p, t in recvs]
            ops += [dist.P2POp(dist.isend, t, p) for p, t in sends]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.staged:
            for _, t, h in landing:
                t.copy_(h)

    def allreduce_sum(self, arr):
        t = self.torch.as_tensor(np.ascontiguousarray(arr)).to(self.wire)
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def allreduce_max(self, x):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.wire)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        self.dist.barrier()

    def broadcast_bytes(self, data):
        """rank 0's bytes on every rank (the RCCL unique id)."""
        n = self.torch.tensor([len(data) if data is not None else 0], dtype=self.torch.int64, device=self.wire)
        self.dist.broadcast(n, 0)
        buf = (self.torch.frombuffer(bytearray(data), dtype=self.torch.uint8).to(self.wire) if data is not None
               else self.torch.zeros(int(n.item()), dtype=self.torch.uint8, device=self.wire))
        self.dist.broadcast(buf, 0)
        return bytes(buf.cpu().numpy().tobytes())


class LocalComm:
    """In-process ranks (one thread each): lets several slabs share ONE GPU, or CPU test engines run
    without a process group.  Same semantics as TorchDistComm."""

    class Hub:
        def __init__(self, world):
            self.world = world
            self.q = {(s, d): queue.Queue() for s in range(world) for d in range(world)}
            self.bar = threading.Barrier(world)
            self.lock = threading.Lock()
            self.acc = {}

    def __init__(self, hub, rank):
        self.hub, self.rank, self.world = hub, rank, hub.world

    def exchange(self, sends, recvs):
        done = []
        for p, t in sends:
            if t.numel():
                ev = threading.Event()
                self.hub.q[(self.rank, p)].put((t, ev))
                done.append(ev)
        for p, t in recvs:
            if t.numel():
                src, ev = self.hub.q[(p, self.rank)].get(timeout=120)
                assert src.numel() == t.numel(), (src.shape, t.shape)
                t.copy_(src.reshape(t.shape))
                ev.set()
        for ev in done:                        # a sender's buffer stays untouched until it was copied
            if not ev.wait(timeout=120):
                raise TimeoutError("LocalComm: a message was never received")

    def _allreduce(self, key, value, fn):
        with self.hub.lock:
            self.hub.acc.setdefault(key, []).append(value)
        self.hub.bar.wait()
        out = fn(self.hub.acc[key])
        self.hub.bar.wait()
        if self.rank == 0:
            self.hub.acc.pop(key, None)
        self.hub.bar.wait()
        return out

    def allreduce_sum(self, arr):
        return self._allreduce("sum", np.asarray(arr), lambda xs: np.sum(xs, axis=0))

    def allreduce_max(self, x):
        return self._allreduce("max", float(x), max)

    def barrier(self):
        self.hub.bar.wait()


# ------------------------------------------------------------------------------------------------
# product engine: libsph_hip.so
# ------------------------------------------------------------------------------------------------
class HipEngine:
    """One z-slab on one MI355X through the C ABI.  Buffers handed to the halo calls are torch
    CUDA tensors (plumbing: device memory + RCCL); all arithmetic is in the HIP library."""

    def __init__(self, capacity, ghost_capacity, params, z_lo, z_hi, device_index=0, ghost_layers=1):
        import sys
        if capi._lib is not None and not capi._torch_first and "torch" not in sys.modules:
            raise capi.SphError("libsph_hip.so was loaded before torch was imported: import torch first "
                                "(see gpufluidsimulator_amd.capi.load)")
        import torch
        n_dev, is950 = capi.device_count()
        if n_dev <= 0 or not is950:
            raise capi.SphError("HipEngine needs a gfx950 (MI355X) device: there is no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.ctx = capi.Context(capacity, params=params, device=device_index, slab=(z_lo, z_hi),
                                ghost_capacity=ghost_capacity, ghost_layers=ghost_layers)
        self.ghost_capacity = ghost_capacity

    def buffer(self, rows, cols):
        return self.torch.zeros((rows, cols), dtype=self.torch.float32, device=self.device)

    def small(self, values):
        return self.torch.tensor(values, dtype=self.torch.int64, device=self.device)

    @staticmethod
    def ptr(t):
        return t.data_ptr()

    @property
    def n(self): return self.ctx.n
    def upload(self, pos, vel, index): self.ctx.upload(pos, vel, index)
    def reset_lattice(self, lattice, jitter, jitter_dims, start, count):
        self.ctx.reset_lattice(lattice, jitter=jitter, jitter_dims=jitter_dims, start=start, count=count)
    def hash(self): self.ctx.hash()
    def sort(self): self.ctx.sort()
    def sort_skipped(self): return self.ctx.sort_skipped()
    def build_cells(self): self.ctx.build_cells()
    def density(self): self.ctx.density()
    def slab_counts(self): return self.ctx.slab_counts()
    def migrants_pack(self, lo, hi): self.ctx.migrants_pack(self.ptr(lo), self.ptr(hi), lo.shape[0])
    def migrants_append(self, buf, n): self.ctx.migrants_append(self.ptr(buf), n)
    def halo_pack(self, lo, hi, counts=None): self.ctx.halo_pack(self.ptr(lo), self.ptr(hi), lo.shape[0], counts)
    def halo_unpack(self, lo, n_lo, hi, n_hi): self.ctx.halo_unpack(self.ptr(lo), n_lo, self.ptr(hi), n_hi)
    def halo_pack_density(self, lo, hi): self.ctx.halo_pack_density(self.ptr(lo), self.ptr(hi), lo.shape[0])
    def halo_unpack_density(self, lo, hi): self.ctx.halo_unpack_density(self.ptr(lo), self.ptr(hi))

    def force_collide_integrate(self, dt): self.ctx.force_collide_integrate(dt)

    def sync(self): self.ctx.sync()

    def download(self, total):
        return self.ctx.download(index_base=0, count=total)

    def download_owned(self): return self.ctx.download_owned()
    def layer_histogram(self): return self.ctx.layer_histogram()

    def to_device(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)

    def close(self): self.ctx.close()


# ------------------------------------------------------------------------------------------------
# the per-rank driver
# ------------------------------------------------------------------------------------------------
class SlabSimulation:
    def __init__(self, comm, engine_factory, box, grid, lattice=None, jitter=True, jitter_dims=None,
                 capacity_factor=1.5, ghost_factor=3.0, particles=None, capacity_slack=4096, device_lattice=None, min_layers=None,
                 python_protocol=3):
        """comm: TorchDistComm | LocalComm.  engine_factory(capacity, ghost_capacity, params, z_lo, z_hi)
        builds this rank's engine.  Either `lattice` (dam break generated slab by slab) or
        `particles` = (pos, vel) of the WHOLE system (small tests)."""
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world
        try:                       # torch before libsph_hip.so (capi.load): one HIP runtime per process
            import torch  # noqa: F401
        except ImportError:
            pass
        self.box = tuple(float(b) for b in box)
        self.grid = tuple(int(g) for g in grid)
        self.params = capi.default_params(self.box, self.grid)
        # which protocol THIS class's Python step speaks (NativeSlabSimulation's step is the library's: its own `protocol`)
        self.python_protocol = int(python_protocol)
        if self.python_protocol == 1:
            min_layers = max(int(min_layers or 0), 4)
            ghost_factor = max(ghost_factor, 6.0)
        self._one_prev = None            # [[sent lo, sent hi], [received lo, received hi]] record totals of the last step
        self.min_layers = int(min_layers or MIN_SLAB_LAYERS)
        gz = self.grid[2]
        if device_lattice is None:         # the product engine generates its lattice layers on the device when it can
            device_lattice = getattr(engine_factory, "device_lattice", False)
        device_run = None                  # (first creation index, count) when the engine makes the particles itself

        if particles is not None:
            pos_all, vel_all = particles
            self.total = int(pos_all.shape[0])
            layers = cell_layer_of(pos_all[:, 2], self.box[2], gz)
            hist = np.bincount(layers, minlength=gz)
            self.cuts = choose_cuts(hist, self.world, self.min_layers)
            z_lo, z_hi = self.cuts[self.rank], self.cuts[self.rank + 1]
            mine = np.nonzero((layers >= z_lo) & (layers < z_hi))[0]
            pos, vel, index = pos_all[mine], vel_all[mine], mine.astype(np.uint32)
        else:
            nx, ny, nz = (int(v) for v in lattice)
            self.total = nx * ny * nz
            jd = jitter_dims if jitter_dims is not None else self.box
            # per-layer histogram from the z coordinates of one lattice column per z (x,y do not matter
            # for z jitter? they do: the jitter is per particle) -> exact: every rank hashes its share
            share = -(-nz // self.world)
            iz0, iz1 = min(self.rank * share, nz), min((self.rank + 1) * share, nz)
            hist = np.zeros(gz, dtype=np.int64)
            for iz in range(iz0, iz1):
                p, _ = ic.dam_break_lattice((nx, ny, nz), self.box, jitter, start=iz * nx * ny, count=nx * ny,
                                            jitter_dims=jd)
                hist += np.bincount(cell_layer_of(p[:, 2], self.box[2], gz), minlength=gz)
            hist = comm.allreduce_sum(hist).astype(np.int64)
            self.cuts = choose_cuts(hist, self.world, self.min_layers)
            z_lo, z_hi = self.cuts[self.rank], self.cuts[self.rank + 1]
            # lattice layers that can reach this slab: every layer whose particles hashed into it
            spacing, radius = float(ic.SPACING), float(ic.PARTICLE_RADIUS)
            amp = 0.5 * float(jd[2]) * 0.01 * radius if jitter else 0.0
            zc = -self.box[2] / 2 + radius + spacing * np.arange(nz)
            lo_l = cell_layer_of(zc - amp - 1e-4, self.box[2], gz)
            hi_l = cell_layer_of(zc + amp + 1e-4, self.box[2], gz)
            cand = np.nonzero((hi_l >= z_lo) & (lo_l < z_hi))[0]
            # Whole lattice layers, none of them straddling a cut (every BASELINE config: the lattice planes sit a quarter
            # of a cell from the faces, the jitter is 0.08 of a cell): the slab's particles are ONE run of creation
            # indices, which the product engine generates in HBM itself (sph_reset_lattice, bit-identical to ic.py) --
            # no host arrays, no PCIe; 8.4 M particles per rank at config 4 otherwise take numpy tens of seconds.
            if (device_lattice and cand.size and np.all(np.diff(cand) == 1) and np.all(lo_l[cand] >= z_lo)
                    and np.all(hi_l[cand] < z_hi)):
                device_run = (int(cand[0]) * nx * ny, int(cand.size) * nx * ny)
                cand = cand[:0]
            chunks_p, chunks_i = [], []
            for iz in cand:
                start = int(iz) * nx * ny
                p, _ = ic.dam_break_lattice((nx, ny, nz), self.box, jitter, start=start, count=nx * ny, jitter_dims=jd)
                lay = cell_layer_of(p[:, 2], self.box[2], gz)
                keep = (lay >= z_lo) & (lay < z_hi)
                chunks_p.append(p[keep])
                chunks_i.append((start + np.nonzero(keep)[0]).astype(np.uint32))
            pos = np.concatenate(chunks_p) if chunks_p else np.zeros((0, 3), np.float32)
            index = np.concatenate(chunks_i) if chunks_i else np.zeros((0,), np.uint32)
            vel = np.zeros_like(pos)
        self.z_lo, self.z_hi = z_lo, z_hi
        n_own = int(pos.shape[0]) if device_run is None else device_run[1]
        per_layer = max(int(hist.max()), 1)
        self.per_layer = per_layer       # particles of the fullest cell layer: the size of a halo message (same on every rank)
        self.ghost_capacity = int(ghost_factor * per_layer) + 1024
        self.capacity = int(capacity_factor * max(n_own, self.total // self.world)) + int(capacity_slack)
        self._factory = engine_factory
        self.engine = engine_factory(self.capacity, self.ghost_capacity, self.params, z_lo, z_hi)
        if self.python_protocol == 1 and not hasattr(self.engine, "pack_one"):
            raise ValueError("python_protocol=1 needs an engine with two ghost layers and pack_one / second_layer_counts (the product "
                             "path runs this protocol inside the library: NativeSlabSimulation(protocol=1))")
        if device_run is None:
            self.engine.upload(pos, vel, index)
        else:
            self.engine.reset_lattice((nx, ny, nz), jitter, jd, device_run[0], device_run[1])
        self._alloc_buffers()
        self.lo_peer = self.rank - 1 if self.rank > 0 else None
        self.hi_peer = self.rank + 1 if self.rank + 1 < self.world else None
        self.stats = {"migrants": 0, "resorts": 0, "ghosts": 0}
        self._counts = None          # slab counts of the previous step, reusable while the sort finds nothing to do

    def _alloc_buffers(self):
        e, g = self.engine, self.ghost_capacity
        self.send_lo, self.send_hi = e.buffer(g, REC), e.buffer(g, REC)
        self.recv_lo, self.recv_hi = e.buffer(g, REC), e.buffer(g, REC)
        self.dsend_lo, self.dsend_hi = e.buffer(g, 2), e.buffer(g, 2)
        self.drecv_lo, self.drecv_hi = e.buffer(g, 2), e.buffer(g, 2)
        self.cnt_recv_lo, self.cnt_recv_hi = e.small([0, 0, 0]), e.small([0, 0, 0])

    # -- one time step ---------------------------------------------------------------------------------
    @staticmethod
    def one_message_rows(prev_total):
        """Records the one message of a step carries behind its header, from the count both ends saw in the step before
        (csrc/sph_slab.hip: one_cap)."""
        return (int(prev_total) + int(prev_total) // 16 + 1024 + 63) & ~63

    def _step_one(self, dt):
        """The ONE-MESSAGE step in Python (the statement of csrc/sph_slab.hip's protocol 1 that CPU ranks can run): header,
        leavers and the residents of the two layers next to each cut in one message per neighbour whose size was fixed from
        the previous step's counts; the receiver completes its copy of the neighbour's layers with its own leavers and
        computes the ghost densities itself (the engine's density pass covers owned + ghosts: the inner ghost layer has its
        whole neighbourhood here).  No density message.  A message that outgrew its size sends the rest in a second one."""
        import torch
        e, c = self.engine, self.comm
        peers = (self.lo_peer, self.hi_peer)
        e.hash(); e.sort()
        m_lo, own_lo, own_hi, m_hi = e.slab_counts()
        n2_lo, n2_hi = e.second_layer_counts()
        if peers[0] is None: assert m_lo == 0, "particles below the box floor"
        if peers[1] is None: assert m_hi == 0, "particles above the box ceiling"
        mine = ((m_lo, own_lo, n2_lo), (m_hi, own_hi, n2_hi))
        tot_s = [sum(mine[k]) if peers[k] is not None else 0 for k in (0, 1)]
        S_s = [self.one_message_rows(self._one_prev[0][k]) if peers[k] is not None else 0 for k in (0, 1)]
        S_r = [self.one_message_rows(self._one_prev[1][k]) if peers[k] is not None else 0 for k in (0, 1)]
        g = self.ghost_capacity
        assert max(S_s + S_r + tot_s) + 1 <= g, "the one message exceeds the buffers"
        send, recv = (self.send_lo, self.send_hi), (self.recv_lo, self.recv_hi)
        e.pack_one(send[0], send[1])                       # rows 1..: leavers, then two layers of residents; leavers dropped
        for k in (0, 1):
            send[k][0, :3] = e.small(list(mine[k])).to(send[k].dtype)       # (counts < 2^24: exact in fp32)
        sends = [(peers[k], send[k][:1 + S_s[k]]) for k in (0, 1) if peers[k] is not None]
        recvs = [(peers[k], recv[k][:1 + S_r[k]]) for k in (0, 1) if peers[k] is not None]
        c.exchange(sends, recvs)
        theirs = [tuple(int(v) for v in recv[k][0, :3].tolist()) if peers[k] is not None else (0, 0, 0) for k in (0, 1)]
        tot_r = [sum(theirs[k]) for k in (0, 1)]
        assert max(tot_r) + 1 <= g, "a neighbour's message exceeds the buffers"
        if any(tot_s[k] > S_s[k] or tot_r[k] > S_r[k] for k in (0, 1)):         # a burst: the rest, exact (both ends know both totals)
            sends = [(peers[k], send[k][1 + S_s[k]:1 + tot_s[k]]) for k in (0, 1) if peers[k] is not None and tot_s[k] > S_s[k]]
            recvs = [(peers[k], recv[k][1 + S_r[k]:1 + tot_r[k]]) for k in (0, 1) if peers[k] is not None and tot_r[k] > S_r[k]]
            c.exchange(sends, recvs)
            self.stats["one_rests"] = self.stats.get("one_rests", 0) + 1
        # arrivals join the owned set; a second sort merges them (as in the three-group step)
        arrived = False
        for k in (0, 1):
            if theirs[k][0]:
                e.migrants_append(recv[k][1:1 + theirs[k][0]], theirs[k][0]); arrived = True
        if arrived:
            e.hash(); e.sort()
            self.stats["resorts"] += 1
        self.stats["migrants"] += m_lo + m_hi
        # ghosts of a side: the neighbour's two layers of residents + what I just sent into them
        ghosts = []
        for k in (0, 1):
            res = recv[k][1 + theirs[k][0]:1 + tot_r[k]]
            lv = send[k][1:1 + mine[k][0]]
            ghosts.append(torch.cat([res, lv]))
        e.halo_unpack(ghosts[0], ghosts[0].shape[0], ghosts[1], ghosts[1].shape[0])
        self.stats["ghosts"] += ghosts[0].shape[0] + ghosts[1].shape[0]
        e.build_cells()
        e.density()
        e.force_collide_integrate(dt)
        self._one_prev = [tot_s, tot_r]
        self.stats["one_steps"] = self.stats.get("one_steps", 0) + 1

    def step(self, dt):
        if self.python_protocol == 1 and self._one_prev is not None:
            return self._step_one(dt)
        e, c = self.engine, self.comm
        lo, hi = self.lo_peer, self.hi_peer
        e.hash(); e.sort()
        if self._counts is not None and e.sort_skipped():
            m_lo, own_lo, own_hi, m_hi = self._counts        # nobody changed cell: same boundary layers, nobody left
        else:
            m_lo, own_lo, own_hi, m_hi = self._counts = e.slab_counts()
        if lo is None: assert m_lo == 0, "particles below the box floor"
        if hi is None: assert m_hi == 0, "particles above the box ceiling"
        # counts: {migrants towards the peer, my boundary-layer particles that stay, my SECOND layer's (what a following
        # one-message step sizes its message from)}
        n2_lo, n2_hi = e.second_layer_counts() if self.python_protocol == 1 else (0, 0)
        sends, recvs = [], []
        if lo is not None:
            sends.append((lo, e.small([m_lo, own_lo, n2_lo]))); recvs.append((lo, self.cnt_recv_lo))
        if hi is not None:
            sends.append((hi, e.small([m_hi, own_hi, n2_hi]))); recvs.append((hi, self.cnt_recv_hi))
        c.exchange(sends, recvs)
        in_lo, peer_own_lo, n2p_lo = (int(v) for v in self.cnt_recv_lo.tolist()) if lo is not None else (0, 0, 0)
        in_hi, peer_own_hi, n2p_hi = (int(v) for v in self.cnt_recv_hi.tolist()) if hi is not None else (0, 0, 0)
        if self.python_protocol == 1:      # both ends of a link hold the same six numbers
            self._one_prev = [[m_lo + own_lo + n2_lo if lo is not None else 0, m_hi + own_hi + n2_hi if hi is not None else 0],
                              [in_lo + peer_own_lo + n2p_lo, in_hi + peer_own_hi + n2p_hi]]
        g = self.ghost_capacity
        assert max(m_lo, m_hi, in_lo, in_hi) <= g, "migrant burst exceeds the ghost capacity"
        # migrants
        if m_lo or m_hi or in_lo or in_hi:
            e.migrants_pack(self.send_lo, self.send_hi)
            sends, recvs = [], []
            if lo is not None:
                sends.append((lo, self.send_lo[:m_lo])); recvs.append((lo, self.recv_lo[:in_lo]))
            if hi is not None:
                sends.append((hi, self.send_hi[:m_hi])); recvs.append((hi, self.recv_hi[:in_hi]))
            c.exchange(sends, recvs)
            if in_lo: e.migrants_append(self.recv_lo, in_lo)
            if in_hi: e.migrants_append(self.recv_hi, in_hi)
            if in_lo or in_hi:
                e.hash(); e.sort()               # newcomers are merged by a second sort (rare, small)
                self.stats["resorts"] += 1
            self.stats["migrants"] += m_lo + m_hi
            self._counts = None                  # the owned set changed: count again next step
        # halo A: boundary layers.  What I send = what stayed in my boundary layer + what arrived in it;
        # the peer knows both numbers (it sent me the second one), so no further count message.
        h_lo, h_hi = own_lo + in_lo, own_hi + in_hi
        gl, gh = peer_own_lo + m_lo, peer_own_hi + m_hi     # ghosts I receive
        assert max(h_lo, h_hi, gl, gh) <= g, "boundary layer exceeds the ghost capacity"
        e.halo_pack(self.send_lo, self.send_hi, (h_lo, h_hi))
        sends, recvs = [], []
        if lo is not None:
            sends.append((lo, self.send_lo[:h_lo])); recvs.append((lo, self.recv_lo[:gl]))
        if hi is not None:
            sends.append((hi, self.send_hi[:h_hi])); recvs.append((hi, self.recv_hi[:gh]))
        c.exchange(sends, recvs)
        e.halo_unpack(self.recv_lo, gl if lo is not None else 0, self.recv_hi, gh if hi is not None else 0)
        self.stats["ghosts"] += gl + gh
        e.build_cells()
        e.density()
        # halo B: density + pressure of the same boundary particles, same order
        e.halo_pack_density(self.dsend_lo, self.dsend_hi)
        sends, recvs = [], []
        if lo is not None:
            sends.append((lo, self.dsend_lo[:h_lo])); recvs.append((lo, self.drecv_lo[:gl]))
        if hi is not None:
            sends.append((hi, self.dsend_hi[:h_hi])); recvs.append((hi, self.drecv_hi[:gh]))
        c.exchange(sends, recvs)
        e.halo_unpack_density(self.drecv_lo, self.drecv_hi)
        e.force_collide_integrate(dt)

    def run(self, dt, steps, rebalance_every=0):
        for k in range(steps):
            self.step(dt)
            if rebalance_every and (k + 1) % rebalance_every == 0:
                self.rebalance()

    # -- re-cut the slabs ("cuts chosen so particle counts are equal ... re-cut every K steps") --
    def _layer_histogram(self):
        """Global per-layer particle histogram (collective).  The product engine counts on the device
        (sph_layer_histogram: 4 bytes per layer cross the bus); the CPU test engine bins its host arrays."""
        e, gz = self.engine, self.grid[2]
        if hasattr(e, "layer_histogram"):
            local = np.asarray(e.layer_histogram(), dtype=np.int64)
        else:
            pos, _, _ = e.download_owned()
            layers = cell_layer_of(pos[:, 2], self.box[2], gz) if pos.shape[0] else np.zeros(0, np.int64)
            local = np.bincount(layers, minlength=gz).astype(np.int64)
        return self.comm.allreduce_sum(local).astype(np.int64)

    def plan_rebalance(self, tolerance=0.02):
        """The cuts a re-balancing would choose now, or None while the most loaded rank is within `tolerance` of the
        mean (collective, cheap: one small all-reduce, nothing moves)."""
        hist = self._layer_histogram()
        counts = [int(hist[a:b].sum()) for a, b in zip(self.cuts, self.cuts[1:])]
        if max(counts) <= (1.0 + tolerance) * self.total / self.world:
            return None                                    # still balanced: keep the cuts
        cuts = choose_cuts(hist, self.world, self.min_layers)
        return None if cuts == self.cuts else cuts

    def rebalance(self, tolerance=0.02, cuts=None):
        """New count-balanced cuts from the current particle distribution; particles whose layer changed
        owner travel point to point.  Collective: every rank must call it.  Returns True if the cuts moved."""
        e, c = self.engine, self.comm
        gz = self.grid[2]
        if cuts is None:
            cuts = self.plan_rebalance(tolerance)
        if cuts is None:
            return False
        pos, vel, idx = e.download_owned()
        layers = cell_layer_of(pos[:, 2], self.box[2], gz) if pos.shape[0] else np.zeros(0, np.int64)
        dest = np.searchsorted(np.asarray(cuts[1:-1]), layers, side="right")    # rank that owns each layer now
        rec = np.zeros((pos.shape[0], REC), np.float32)
        rec[:, 0:3] = pos; rec[:, 3] = idx.view(np.float32); rec[:, 4:7] = vel
        # who sends how many to whom: one world x world matrix, every rank fills its row
        m = np.zeros((self.world, self.world), np.int64)
        m[self.rank] = np.bincount(dest, minlength=self.world)
        m = c.allreduce_sum(m).astype(np.int64)
        sends, recvs, inbox = [], [], {}
        for peer in range(self.world):
            if peer == self.rank:
                continue
            if m[self.rank, peer]:
                sends.append((peer, e.to_device(rec[dest == peer])))
            if m[peer, self.rank]:
                inbox[peer] = e.buffer(int(m[peer, self.rank]), REC)
                recvs.append((peer, inbox[peer]))
        c.exchange(sends, recvs)
        # Upload order = the order of the whole-domain sorted array restricted to my new layers: what the ranks below
        # me held (each in its slot order), what I keep, what the ranks above me held.  The new context's first sort
        # is a stable radix sort, so every cell keeps the order it has in a one-context run (bit-identical sums).
        parts = ([inbox[p].cpu().numpy() for p in sorted(inbox) if p < self.rank] + [rec[dest == self.rank]] +
                 [inbox[p].cpu().numpy() for p in sorted(inbox) if p > self.rank])
        rec = np.concatenate(parts) if parts else np.zeros((0, REC), np.float32)
        # a fresh engine for the new layer range (cell table, key width and capacities depend on it)
        e.close()
        self.cuts = cuts
        self.z_lo, self.z_hi = cuts[self.rank], cuts[self.rank + 1]
        n_own = rec.shape[0]
        self.capacity = max(self.capacity, int(1.5 * n_own) + 4096)
        self.engine = self._factory(self.capacity, self.ghost_capacity, self.params, self.z_lo, self.z_hi)
        self._counts = None
        self._one_prev = None              # (new layers: the next step learns the one message's sizes again)
        self.engine.upload(rec[:, 0:3].copy(), rec[:, 4:7].copy(), np.ascontiguousarray(rec[:, 3]).view(np.uint32).copy())
        self._alloc_buffers()
        self.stats["rebalances"] = self.stats.get("rebalances", 0) + 1
        return True

    def gather_state(self):
        """Every rank's owned particles by creation index (NaN elsewhere) -> combined on all ranks."""
        st = self.engine.download(self.total)
        out = {}
        for k, v in st.items():
            filled = np.where(np.isnan(v), 0.0, v).astype(np.float64)
            have = (~np.isnan(v)).astype(np.float64)
            s = self.comm.allreduce_sum(filled)
            h = self.comm.allreduce_sum(have)
            assert np.all(h == 1.0), f"{k}: {int((h != 1.0).sum())} entries owned by != 1 rank"
            out[k] = s.astype(np.float32)
        return out


# ------------------------------------------------------------------------------------------------
# the native step: csrc/sph_slab.hip under the C ABI (sph_slab_step); Python is launcher and set-up only
# ------------------------------------------------------------------------------------------------
def slab_timing_dict(slab_handle):
    """sph_slab_timing_get of an `sph_slab*` as a dict (microseconds): host_wait / host_pre / host_post / host_step {mean, max}
    over the steps since the last reset, waits_ready (the header was there before the host looked: host-paced steps), and per
    message group {calls, mean, max} of the event pairs on the comm stream (only while sph_slab_timing_enable)."""
    import ctypes as C
    w = (C.c_double * 28)()
    capi._check(capi.load().sph_slab_timing_get(slab_handle, w))
    n = max(w[0], 1.0)
    out = {"steps": int(w[0]), "waits_ready": int(w[1])}
    for k, name in enumerate(("host_wait_us", "host_pre_us", "host_post_us", "host_step_us")):
        out[name] = {"mean": w[2 + 2 * k] / n, "max": w[3 + 2 * k]}
    for g, name in enumerate(_GROUPS):       # the three-group protocol's four tags, then the one-message protocol's two
        at = 10 + 3 * g
        c = w[at]
        out["exchange_us_" + name] = {"calls": int(c), "mean": w[at + 1] / c if c else None, "max": w[at + 2]}
    return out


def host_transport(comm):
    """A `sph_transport` that moves the library's pinned HOST staging buffers through `comm.exchange`
    (LocalComm: several slabs of one GPU in one process; TorchDistComm over gloo: several processes on one
    GPU).  For rehearsals and tests -- the product transport is RCCL (rccl_transport)."""
    import ctypes as C

    import torch

    def view(ptr, nbytes):
        return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))) if nbytes else None

    lo = comm.rank - 1 if comm.rank > 0 else None
    hi = comm.rank + 1 if comm.rank + 1 < comm.world else None

    def exchange(_self, _tag, send_lo, n_send_lo, recv_lo, n_recv_lo, send_hi, n_send_hi, recv_hi, n_recv_hi, _stream):
        try:
            sends, recvs = [], []
            if lo is not None:
                if n_send_lo: sends.append((lo, view(send_lo, n_send_lo)))
                if n_recv_lo: recvs.append((lo, view(recv_lo, n_recv_lo)))
            if hi is not None:
                if n_send_hi: sends.append((hi, view(send_hi, n_send_hi)))
                if n_recv_hi: recvs.append((hi, view(recv_hi, n_recv_hi)))
            comm.exchange(sends, recvs)
            return 0
        except BaseException as e:     # noqa: BLE001 -- an exception must not unwind through the C caller
            import traceback
            traceback.print_exc()
            host_transport.last_error = e
            return -2

    fn = capi.EXCHANGE_FN(exchange)
    t = capi.Transport(None, fn, 1)
    t._keepalive = fn
    return t


def rccl_transport(rank, world, device_index, broadcast_id):
    """The product transport: RCCL send/recv on the library's comm stream.  `broadcast_id(bytes|None) -> bytes`
    hands rank 0's 128-byte id to every rank (the launcher's job: torch.distributed broadcast, a file, ...)."""
    import ctypes as C
    L = capi.load()
    buf = (C.c_uint8 * 128)()
    if rank == 0:
        capi._check(L.sph_rccl_unique_id(buf))
    raw = broadcast_id(bytes(buf) if rank == 0 else None)
    buf = (C.c_uint8 * 128).from_buffer_copy(raw)
    t = C.POINTER(capi.Transport)()
    capi._check(L.sph_rccl_transport_create(C.byref(t), buf, rank, world, device_index))
    return t


class NativeSlabSimulation(SlabSimulation):
    """Set-up (cuts, slab-by-slab lattice, upload, re-balancing, gathering) as SlabSimulation; the time step is ONE
    call into libsph_hip.so per rank: sph_slab_step queues sort, migrants, halo A, density, halo B and the fused
    force pass on two HIP streams and waits for the device once (csrc/sph_slab.hip)."""

    # the early force launch (sph_slab_set_early_force) costs 5-12 us per step when a message group takes less than this on an
    # idle device (the preflight ping of the 8 KB migrant message) and wins beyond it: measured on a slab between its periodic
    # images with the round's last code (profiles/r06d_periodic_slab_protocols.txt: at 20 us per group --
    # a ping of 28 -- still 5 us slower sustained, at 40 us -- a ping of 48 -- 30 us faster).  "auto" switches it off for such links.
    EARLY_FORCE_MIN_PING_US = 32.0
    EARLY_FORCE_MIN_HALO_PING_US = 75.0      # ... unless a halo-A-sized message takes more than this (a slow link rather than a late one: 10 us + 75 GB/s is level)
    # ... and off again when the slab is so big that the deep density launch (queued in front of the wait anyway) outlasts the
    # two messages on the path: ~half a slab's particles are "deep", k_density does ~18,800 of them per us (C3, flowing).  A
    # 16.7 M-particle slab (config 5's rank) behind a 10 us / 153 GB/s link: 3.71 ms per step without, 3.85 with it.
    EARLY_FORCE_DEEP_PARTICLES_PER_US = 18.8e3
    EARLY_FORCE_SLACK_US = 20.0

    @classmethod
    def early_force_rule(cls, migrants_us, halo_a_us, owned):
        """(on, why) from the preflight pings of the migrant- and the halo-A-sized message and the rank's particle count."""
        slow = migrants_us >= cls.EARLY_FORCE_MIN_PING_US or halo_a_us >= cls.EARLY_FORCE_MIN_HALO_PING_US
        cover = 0.5 * owned / cls.EARLY_FORCE_DEEP_PARTICLES_PER_US
        exposed = migrants_us + halo_a_us + cls.EARLY_FORCE_SLACK_US > cover
        why = (f"pings: migrant message {migrants_us:.1f} us (on from {cls.EARLY_FORCE_MIN_PING_US:.0f}), halo A {halo_a_us:.1f} us "
               f"(on from {cls.EARLY_FORCE_MIN_HALO_PING_US:.0f}); the deep density launch of {owned} particles covers ~{cover:.0f} us "
               f"(on while the two messages + {cls.EARLY_FORCE_SLACK_US:.0f} us exceed it)")
        return bool(slow and exposed), why

    def __init__(self, comm, box, grid, device_index=0, transport="host", migrant_capacity=0, ping_reps=3, early_force="auto",
                 protocol=None, **kw):
        """protocol: 3 = MIGRANTS / HALO A / HALO B (three dependent message groups per step); 1 = the one-message step
        (sph_slab_set_protocol: two ghost layers, ghost densities recomputed locally; slabs of >= 4 cell layers).  The same on
        every rank."""
        if protocol is None:               # (SPH_SLAB_PROTOCOL=1: the default of callers that do not say -- a test switch)
            protocol = int(os.environ.get("SPH_SLAB_PROTOCOL", "3"))
        if protocol not in (1, 3):
            raise ValueError("protocol is 1 (one message per step) or 3")
        self.protocol = int(protocol)
        if self.protocol == 1:
            kw.setdefault("min_layers", 4)
            kw.setdefault("ghost_factor", 5.0)             # two layers of ghosts per side
        self._device_index = device_index
        self._transport_kind = transport
        self._migrant_capacity = migrant_capacity
        self._slab = None
        self._tr = None
        self._ping_reps = int(ping_reps)
        self._early_force = early_force    # True / False / "auto": from the pings (collective: the same decision on every rank)
        self.early_force = None            # {"on": bool, "why": str}
        self.ping = None               # {bytes: {"mean_us", "max_us"}} of the preflight, per message size of a step
        self.rccl = None               # what the RCCL communicator says about itself (sph_rccl_transport_info)
        # (SPH_SLAB_GHOST_LAYERS=2: the three-group protocol on contexts with two ghost layers -- a test of the layer numbering)
        ghost_layers = 2 if self.protocol == 1 or os.environ.get("SPH_SLAB_GHOST_LAYERS") == "2" else 1

        def factory(cap, gcap, p, z0, z1):
            return HipEngine(cap, gcap, p, z0, z1, device_index, ghost_layers)
        factory.device_lattice = True
        super().__init__(comm, factory, box, grid, **kw)
        self._bind()

    def _alloc_buffers(self):          # the library owns the message buffers
        pass

    def _bind(self):
        import ctypes as C
        L = capi.load()
        self._unbind()
        if self._transport_kind == "rccl":
            if self._tr is None:       # one communicator for the life of the run (re-balancing keeps it)
                self._tr = rccl_transport(self.rank, self.world, self._device_index, self.comm.broadcast_bytes)
                # fail here, loudly, not inside a step: with neighbours the preflight ping below is that check (the step's own
                # message pattern, contents verified); a chain of one -- or a caller that switched the ping off -- sends to itself
                if self.world == 1 or self._ping_reps <= 0:
                    capi._check(L.sph_rccl_transport_selftest(self._tr, 1 << 16))
            tr = self._tr
        elif self._transport_kind == "local":
            # device pointers between the streams of this process (csrc/sph_slab.hip: local_exchange); the ranks share
            # one capi.LocalHub, handed over as `comm.local_hub`
            if self._tr is None:
                self._tr = self.comm.local_hub.transport(self.rank)
            tr = self._tr
        else:
            self._tr = host_transport(self.comm)
            tr = C.pointer(self._tr)
        h = C.c_void_p()
        capi._check(L.sph_slab_create(C.byref(h), self.engine.ctx.h, self.rank, self.world, tr, int(self._migrant_capacity)))
        self._slab = h
        if self.protocol == 1:
            capi._check(L.sph_slab_set_protocol(self._slab, 1))
        if self._transport_kind == "rccl" and self.rccl is None:
            info = (C.c_int * 4)()
            capi._check(L.sph_rccl_transport_info(self._tr, info))
            self.rccl = {"world_seen": int(info[0]), "rank_seen": int(info[1]), "device_seen": int(info[2]),
                         "async_error": int(info[3]), "device": int(self._device_index)}
            if self.rccl["world_seen"] not in (-1, self.world) or self.rccl["rank_seen"] not in (-1, self.rank):
                raise capi.SphError(f"rank {self.rank}/{self.world}: the RCCL communicator says it is rank "
                                    f"{self.rccl['rank_seen']} of {self.rccl['world_seen']}")
        if self.ping is None and self.world > 1 and self._ping_reps > 0:
            self.ping = self._preflight()
        self._decide_early_force()

    def _decide_early_force(self):
        if self._early_force in (True, False):
            on, why = bool(self._early_force), "set by the caller"
        elif self._transport_kind != "rccl":
            # ranks that SHARE a device (threads over the device-to-device transport, processes over gloo): one rank's early
            # launch then runs beside the other ranks' big kernels all the time and they evict each other's L2 working sets --
            # measured 4.81 against 3.86 ms per step for two 8.4 M-particle slabs on one GPU.  It is meant for one rank per GPU.
            on, why = False, f"transport '{self._transport_kind}': the ranks share a device"
        elif self.ping is None:
            on, why = True, "no ping taken: the library's default"
        else:       # the slowest link any rank saw decides for all (a rank's choice is its own scheduling: nothing has to agree)
            worst = self.comm.allreduce_max(self.ping["migrants"]["mean_us"])
            worst_a = self.comm.allreduce_max(self.ping["halo_a"]["mean_us"])
            fewest = -self.comm.allreduce_max(-float(self.engine.ctx.n))
            on, why = self.early_force_rule(worst, worst_a, int(fewest))
            why = "max over ranks (particles: min): " + why
        self.early_force = {"on": on, "why": why}
        capi._check(capi.load().sph_slab_set_early_force(self._slab, 1 if on else 0))

    def message_sizes(self):
        """Bytes per direction of the three message groups of a usual step: the fixed migrant message (header + 255
        records), halo A (32 B per particle of a boundary layer) and halo B (8 B per particle) -- for the fullest cell
        layer, clamped to the library's buffers.  The same on every rank (the histogram is global)."""
        cap = (self.ghost_capacity + 1) * 32
        return {"migrants": 8192, "halo_a": min(max(self.per_layer, 1) * 32, cap), "halo_b": min(max(self.per_layer, 1) * 8, cap)}

    def _preflight(self):
        """Neighbour ping before the first step (collective): one exchange-shaped group to rank +- 1 at each of the
        step's three message sizes through the product path (sph_slab_ping: the slab's transport, comm stream and
        buffers; every word checked for sender and direction).  An RCCL error, a neighbour that does not answer or a
        wrong word raises here with the library's message -- there is no fall-back to another transport."""
        import ctypes as C
        L = capi.load()
        out = {}
        for name, nbytes in self.message_sizes().items():
            res = (C.c_double * 3)()
            capi._check(L.sph_slab_ping(self._slab, int(nbytes), self._ping_reps, res))
            out[name] = {"bytes": int(nbytes), "mean_us": float(res[0]), "max_us": float(res[1])}
        return out

    def slab_timing(self, reset=False):
        """sph_slab_timing_get as a dict, see slab_timing_dict."""
        out = slab_timing_dict(self._slab)
        if reset:
            capi._check(capi.load().sph_slab_timing_reset(self._slab))
        return out

    def set_early_force(self, on=True):
        """The force pass of the innermost layers in front of the step's wait (sph_slab_set_early_force); same bits either way.
        An integer > 1: on, and that many slots at most (the library's default: 2^20)."""
        self._early_force = bool(on)
        self.early_force = {"on": bool(on), "why": "set by the caller"}
        capi._check(capi.load().sph_slab_set_early_force(self._slab, int(on)))

    def slab_timing_enable(self, on=True):
        capi._check(capi.load().sph_slab_timing_enable(self._slab, 1 if on else 0))

    def slab_timing_reset(self):
        capi._check(capi.load().sph_slab_timing_reset(self._slab))

    def _unbind(self):
        if self._slab:
            capi.load().sph_slab_destroy(self._slab)
            self._slab = None

    def step(self, dt):
        self.run(dt, 1)

    def run(self, dt, steps, rebalance_every=0):
        L = capi.load()
        done = 0
        while done < steps:
            k = steps - done if not rebalance_every else min(steps - done, rebalance_every - done % rebalance_every)
            capi._check(L.sph_slab_step(self._slab, float(dt), int(k)))
            done += k
            if rebalance_every and done % rebalance_every == 0:
                self.rebalance()
        self._pull_stats()

    def _pull_stats(self):
        import ctypes as C
        out = (C.c_uint64 * 8)()
        capi._check(capi.load().sph_slab_counters(self._slab, out))
        base = getattr(self, "_stats_base", {})
        names = ("steps", "migrants", "resorts", "ghosts", "host_waits", "in_place_merges", "far_steps", "rest_messages")
        self.stats.update({k: base.get(k, 0) + int(out[i]) for i, k in enumerate(names)})
        self.stats["exchanges"] = base.get("exchanges", 0) + int(capi.load().sph_slab_exchanges(self._slab))
        ef = (C.c_uint64 * 2)()
        capi._check(capi.load().sph_slab_early_force_stats(self._slab, ef))
        self.stats["early_force_launches"], self.stats["early_force_used"] = int(ef[0]), int(ef[1])
        pr = (C.c_uint64 * 3)()
        capi._check(capi.load().sph_slab_protocol(self._slab, pr))
        self.stats["protocol"], self.stats["one_message_steps"], self.stats["one_message_rests"] = int(pr[0]), int(pr[1]), int(pr[2])

    def sync(self):
        capi._check(capi.load().sph_slab_sync(self._slab))

    def rebalance(self, tolerance=0.02, cuts=None):
        """Re-cut ON THE DEVICE (sph_slab_recut): whole layers change owner point to point through the slab's own
        transport, the context, the slab object and its buffers are kept -- no host round trip of the particles, no new
        context (round 4 moved them through host numpy).  Collective.  A move of a cut past a neighbouring cut is taken
        in several hops (single_hop_cuts)."""
        self.sync()
        hist = None
        if cuts is None:
            hist = self._layer_histogram()                 # device-side histogram + one small all-reduce
            counts = [int(hist[a:b].sum()) for a, b in zip(self.cuts, self.cuts[1:])]
            if max(counts) <= (1.0 + tolerance) * self.total / self.world:
                return False                               # balanced: nothing moves
            cuts = choose_cuts(hist, self.world, self.min_layers)
        cuts = [int(c) for c in cuts]
        if cuts == self.cuts:
            return False
        if hist is None:
            hist = self._layer_histogram()
        L = capi.load()
        while self.cuts != cuts:
            step = single_hop_cuts(self.cuts, cuts)
            # every rank sees every rank's new count: a slab that would overflow stops the re-cut on ALL ranks, before
            # anything has moved (the histogram counts owned particles by their layer: exact up to the step's leavers)
            need = [int(hist[a:b].sum()) for a, b in zip(step, step[1:])]
            caps = self.comm.allreduce_sum(np.eye(self.world, dtype=np.int64)[self.rank] * int(self.capacity)).astype(np.int64)
            over = [r for r in range(self.world) if need[r] + 1024 > caps[r]]
            if over:
                raise capi.SphError(f"re-balancing to the cuts {step} needs {need[over[0]]} particles on rank {over[0]}, "
                                    f"whose capacity is {int(caps[over[0]])}")
            capi._check(L.sph_slab_recut(self._slab, step[self.rank], step[self.rank + 1]))
            self.cuts = step
            self.z_lo, self.z_hi = step[self.rank], step[self.rank + 1]
        self.stats["rebalances"] = self.stats.get("rebalances", 0) + 1
        return True

    def close(self):
        self._unbind()                 # before the context goes: sph_slab_destroy drains the context's stream
        if self._transport_kind == "rccl" and self._tr is not None:
            capi.load().sph_rccl_transport_destroy(self._tr)
        if self._transport_kind == "local" and self._tr is not None:
            capi.load().sph_local_transport_destroy(self._tr)
        self._tr = None
        self.engine.close()


# ------------------------------------------------------------------------------------------------
# bench entry (bench.py --gpus N, one rank per GPU under torch.distributed.run)
# ------------------------------------------------------------------------------------------------
def bench_config(args, world):
    """BASELINE.json's metric is "dam-break 16M particles, 1/2/4/8 MI355X": the SAME 16,777,216-particle dam (config
    3) on N GPUs -- strong scaling, the default.  `--workload C4` = BASELINE config 4 (67,108,864 particles, strong);
    `--scaling weak` = 16.7 M particles PER GPU (the box grows along z)."""
    strong = getattr(args, "scaling", "strong") != "weak"
    if strong:
        name = getattr(args, "workload", "C3") or "C3"
        cfg = dict(ic.CONFIGS[name], jitter_dims=ic.CONFIGS[name]["box"])
        label = f"{name} (strong scaling: the same {cfg['lattice'][0] * cfg['lattice'][1] * cfg['lattice'][2]} particles on every N)"
    else:
        cfg = ic.weak_scaling_config(world)
        label = "C3 per GPU (weak scaling)"
    lat = getattr(args, "lattice", None)
    if lat:                                            # e.g. one rank's share of C3 on 8 GPUs: --lattice 256,256,32
        cfg["lattice"] = tuple(int(v) for v in lat.split(","))
        label += f", lattice overridden to {lat}"
    return cfg, strong, label


def gather_rows(comm, row):
    """Every rank's vector of floats on every rank: a world x K matrix (one small all-reduce; NaN stands for 'none')."""
    row = np.asarray(row, dtype=np.float64)
    m = np.zeros((comm.world, row.shape[0]), np.float64)
    m[comm.rank] = np.where(np.isnan(row), -1e300, row)
    m = np.asarray(comm.allreduce_sum(m), dtype=np.float64)
    return np.where(m <= -1e299, np.nan, m)


_GROUPS = ("migrants", "halo_a", "halo_b", "migrants_rest", "one", "one_rest")
_HOST = ("host_wait_us", "host_pre_us", "host_post_us", "host_step_us")


def bench_diagnostics(sim, comm, phases_ms, host_t, probe_t, n_own):
    """What makes an N-rank bench line self-diagnosing: EVERY rank's phase times, host-side step timing of the timed
    window, event-timed message groups of the probe pass, the preflight pings and what RCCL says about its communicator
    -- gathered with one small all-reduce, laid out per rank and summarised (mean over ranks, max over ranks)."""
    nan = float("nan")
    row = [float(n_own)] + [float(phases_ms.get(k, 0.0)) for k in capi.PHASES]
    for k in _HOST:
        row += [host_t[k]["mean"], host_t[k]["max"]]
    row += [float(host_t["waits_ready"]), float(host_t["steps"])]
    for g in _GROUPS:
        e = probe_t["exchange_us_" + g]
        row += [float(e["calls"]), nan if e["mean"] is None else e["mean"], e["max"]]
    ping = sim.ping or {}
    for g in _GROUPS[:3]:
        e = ping.get(g)
        row += [nan, nan, nan] if e is None else [float(e["bytes"]), e["mean_us"], e["max_us"]]
    r = sim.rccl or {}
    row += [float(r.get(k, -1)) for k in ("world_seen", "rank_seen", "device_seen", "async_error", "device")]
    m = gather_rows(comm, row)
    none = lambda v: None if np.isnan(v) else float(v)          # noqa: E731
    ranks, col = [], 0
    for q in range(comm.world):
        v, col = m[q], 0
        d = {"rank": q, "owned": int(v[0])}
        col = 1
        d["phases_ms"] = {k: float(v[col + i]) for i, k in enumerate(capi.PHASES)}; col += len(capi.PHASES)
        for k in _HOST:
            d[k] = {"mean": float(v[col]), "max": float(v[col + 1])}; col += 2
        d["waits_ready"], d["steps_timed"] = int(v[col]), int(v[col + 1]); col += 2
        d["exchange_us"] = {}
        for g in _GROUPS:
            d["exchange_us"][g] = {"calls": int(v[col]), "mean": none(v[col + 1]), "max": float(v[col + 2])}; col += 3
        d["ping_us"] = {}
        for g in _GROUPS[:3]:
            if not np.isnan(v[col]):
                d["ping_us"][g] = {"bytes": int(v[col]), "mean": float(v[col + 1]), "max": float(v[col + 2])}
            col += 3
        d["rccl"] = {k: int(v[col + i]) for i, k in enumerate(("world_seen", "rank_seen", "device_seen", "async_error", "device"))}
        ranks.append(d)

    def over_ranks(get):
        vals = [get(d) for d in ranks]
        vals = [x for x in vals if x is not None]
        return {"mean": float(np.mean(vals)), "max": float(np.max(vals))} if vals else None

    out = {"ranks": ranks}
    # per message group: mean over ranks of the per-rank mean, and the worst single group seen on any rank
    out["exchange_us"] = {g: {"mean": (over_ranks(lambda d, g=g: d["exchange_us"][g]["mean"]) or {}).get("mean"),
                              "max": max(d["exchange_us"][g]["max"] for d in ranks),
                              "calls_per_rank": ranks[0]["exchange_us"][g]["calls"]} for g in _GROUPS}
    out["host_wait_us"] = {"mean": float(np.mean([d["host_wait_us"]["mean"] for d in ranks])),
                           "max": float(max(d["host_wait_us"]["max"] for d in ranks)),
                           "waits_ready_frac": float(sum(d["waits_ready"] for d in ranks)) / max(sum(d["steps_timed"] for d in ranks), 1)}
    out["host_step_us"] = {k: {"mean": float(np.mean([d[k]["mean"] for d in ranks])), "max": float(max(d[k]["max"] for d in ranks))}
                           for k in _HOST[1:]}
    out["ping_us"] = {g: {"bytes": ranks[0]["ping_us"][g]["bytes"],
                          "mean": float(np.mean([d["ping_us"][g]["mean"] for d in ranks if g in d["ping_us"]])),
                          "max": float(max(d["ping_us"][g]["max"] for d in ranks if g in d["ping_us"]))}
                      for g in _GROUPS[:3] if g in ranks[0]["ping_us"]}
    out["rccl"] = ({"world_seen": [d["rccl"]["world_seen"] for d in ranks], "rank_seen": [d["rccl"]["rank_seen"] for d in ranks],
                    "devices": [d["rccl"]["device_seen"] for d in ranks],
                    "async_error": [d["rccl"]["async_error"] for d in ranks], "ping_us": out["ping_us"]}
                   if sim.rccl is not None else None)
    out["phases_ms_max_over_ranks"] = {k: float(max(d["phases_ms"][k] for d in ranks)) for k in capi.PHASES}
    return out


def bench_rank(comm, local, args, transport, log=None):
    """One rank of the multi-GPU bench (a process under torch.distributed.run, or a thread of the one-GPU rehearsal).
    Returns the JSON record on rank 0, None elsewhere."""
    import torch
    rank, world = comm.rank, comm.world
    log = log or (lambda msg: print(msg, file=sys.stderr, flush=True))
    cfg, strong, label = bench_config(args, world)
    ef = {"on": True, "off": False}.get(getattr(args, "early_force", "auto"), "auto")
    sim = NativeSlabSimulation(comm, cfg["box"], cfg["grid"], device_index=local, transport=transport,
                               lattice=cfg["lattice"], jitter=True, jitter_dims=cfg["jitter_dims"], early_force=ef,
                               protocol=int(getattr(args, "protocol", 3) or 3))
    layers = [b - a for a, b in zip(sim.cuts, sim.cuts[1:])]
    assert min(layers) >= MIN_SLAB_LAYERS or world == 1, sim.cuts
    mixed = getattr(args, "precision", "f32") == "mixed"        # BASELINE config 5's arithmetic
    sim.engine.ctx.set_precision(mixed)
    dt = float(ic.DEFAULT_DT)
    strict_flow = getattr(args, "runup", None) is None          # the default configuration must be a flowing state
    runup = args.runup if not strict_flow else 6000
    every = int(getattr(args, "rebalance_every", 500) or 0)      # run-up only: the timed window keeps its cuts
    tail = min(1000, runup)
    t0 = time.perf_counter()
    left, q0, t_tail = runup, None, None
    while left > 0:                                    # state preparation: the flowing dam (as in the 1-GPU bench)
        if left == tail:
            sim.sync(); comm.barrier()
            q0, t_tail = sim.engine.ctx.sort_stats(), time.perf_counter()
        k = min(left - tail, 1000) if left > tail else min(left, 1000)
        sim.run(dt, k, rebalance_every=every); sim.sync()
        left -= k
        if rank == 0:
            log(f"[bench] run-up {runup - left}/{runup} steps, {time.perf_counter() - t0:.1f} s, cuts {sim.cuts}")
    sim.sync(); comm.barrier()
    tail_wall = comm.allreduce_max(time.perf_counter() - t_tail) if t_tail is not None else 0.0
    q1 = sim.engine.ctx.sort_stats()
    tail_movers = (q1["movers_total"] - q0["movers_total"]) if q0 is not None else 0
    rebalances = sim.stats.get("rebalances", 0)
    sim.run(dt, args.warmup)
    sim.sync(); torch.cuda.synchronize(); comm.barrier()
    s0 = sim.engine.ctx.sort_stats()
    sim.slab_timing_reset()                              # host-side step timing of the timed window (three clock reads a step)
    t0 = time.perf_counter()
    sim.run(dt, args.steps)
    sim.sync(); torch.cuda.synchronize(); comm.barrier()
    wall = comm.allreduce_max(time.perf_counter() - t0)
    host_t = sim.slab_timing()
    s1 = sim.engine.ctx.sort_stats()
    n_own = sim.engine.n
    counts = comm.allreduce_sum(np.array([n_own, s1["movers_total"] - s0["movers_total"], s1["skips"] - s0["skips"],
                                          tail_movers], dtype=np.int64))
    n_max = comm.allreduce_max(n_own)
    # per-phase device times of this rank's kernels (HIP events on the library's stream), outside the timed region
    # -- and event pairs around every message group on the comm stream (sph_slab_timing_enable): the probe pass pays the
    #    events' dispatch cost, the timed window above does not
    ctx = sim.engine.ctx
    ctx.timing(True); ctx.timing_reset()
    sim.slab_timing_reset(); sim.slab_timing_enable(True)
    probe = max(2, min(args.steps, 20))
    sim.run(dt, probe)
    sim.sync(); comm.barrier()
    ph, _ = ctx.timing_get()
    ctx.timing(False)
    probe_t = sim.slab_timing()
    sim.slab_timing_enable(False)
    phases_ms = {k: v / probe for k, v in ph.items()}
    diag = bench_diagnostics(sim, comm, phases_ms, host_t, probe_t, n_own)      # collective: every rank
    # ---- rank 0 completes the line the way the N = 1 bench does, the others wait at the barrier: the HBM-side
    #      traffic of ITS pair kernels -- its slab's state saved and stepped alone by a child of bench.py under `rocprofv3 --pmc` --
    #      and the cpu_baseline probe on this box's host cores (the 262,144-particle dam: BASELINE.md section 3)
    traffic, traffic_note, cpu_base = None, "--no-pmc", None
    if rank == 0:
        try:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench as bench_mod
            shared_card = bool(getattr(args, "one_gpu", False)) and getattr(args, "ranks_as", "") == "processes"
            if shared_card:
                # a rehearsal with the ranks as PROCESSES on one card: the profiler and its child would be two more processes with
                # the GPU open, and a GPU box admits six (world 4 + the caller + these two = 7: the run is killed)
                traffic_note = "not measured: the ranks share ONE GPU as processes (--one-gpu under a launcher) and a box admits six processes on a card"
            elif not getattr(args, "no_pmc", False):
                t_p = time.perf_counter()
                traffic, traffic_note = bench_mod.pmc_traffic(sim.engine.ctx, args, slab=(sim.z_lo, sim.z_hi), device=local)
                log(f"[bench] rank 0 PMC traffic passes: {time.perf_counter() - t_p:.1f} s ({'ok' if traffic else traffic_note})")
            if not getattr(args, "no_cpu", False):
                cpu_base = bench_mod.cpu_baseline(budget_s=float(getattr(args, "cpu_budget", 24.0)))
                cpu_base["sample"] += ("; the probe of the N = 1 bench (no leg on this run's own particles: a rank holds a slab of them), "
                                       "timed on rank 0's host while the other ranks wait")
        except Exception as e:      # noqa: BLE001 -- the line must not depend on the profiler or the oracle build
            traffic_note = f"{type(e).__name__}: {e}"
    comm.barrier()
    out = None
    if rank == 0:
        total = sim.total
        movers_win = float(counts[1]) / max(args.steps, 1)
        movers_tail = float(counts[3]) / max(tail, 1)
        # the N = 1 bench's test (bench.py): no sort of the window skipped, particles changed cell in it, and over the
        # last 1000 run-up steps at least 1e-3 N of them did so per step
        flow_ok = bool(int(counts[2]) == 0 and movers_win >= 1e-4 * total and movers_tail >= 1e-3 * total)
        t_force = max(phases_ms.get("force", 0.0), 1e-9) * 1e-3
        out = {
            "metric": "particle-steps/sec", "value": total * args.steps / wall, "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32 state + packed-f16 density pairs (config 5's arithmetic: an option, slower than fp32 on gfx950)" if mixed else "f32",
            "data": "synthetic",
            "config": {"workload": f"dam-break {label}: "
                                   f"{cfg['lattice'][0]}x{cfg['lattice'][1]}x{cfg['lattice'][2]} = {total} particles "
                                   f"({total // world} per GPU), grid {list(cfg['grid'])}, dt 5e-7, "
                                   f"{'FLOWING' if flow_ok else 'NOT a flowing state'}: timed after {runup} run-up steps; z-slabs, "
                                   + ("ONE slab, NO neighbours: nothing is exchanged (no halo pack / unpack, no boundary launches)"
                                      if world == 1 else "ghost layers and migrants over "
                                      + {"rccl": "RCCL send/recv (library comm stream)", "host": "host-staged messages",
                                         "local": "device-to-device copies between the streams of one process"}[transport]),
                       "particles": total, "grid": list(cfg["grid"]), "cuts": sim.cuts, "layers_per_slab": layers,
                       "state": "flow" if flow_ok else "not-flowing", "runup_steps": runup,
                       "runup_last_1000_steps": {"steps": tail, "movers_per_step": movers_tail, "rebalances": rebalances,
                                                 "ms_per_step": tail_wall / max(tail, 1) * 1e3},
                       "parallelism": f"{world} z-slabs, one per GPU", "transport": transport,
                       "ranks_as": getattr(args, "ranks_as", "processes")},
            # the sustained figure: the last 1000 run-up steps between two syncs (re-balancing included)
            "value_sustained": total * tail / tail_wall if tail_wall > 0 else None,
            "ms_per_step_sustained": tail_wall / max(tail, 1) * 1e3 if tail_wall > 0 else None,
            # rank 0's fused force pass (its interior + boundary launches of one step), algorithmic bytes as at N = 1
            "roofline": {"bound": "valu-issue", "kernel": "k_force<force+collision+integrate> (rank 0, launches of one step)",
                         "achieved": 84.0 * n_own / t_force / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": 84.0 * n_own / t_force / 1e9 / 8000.0,
                         "traffic": traffic["force_fused"] if traffic else None,
                         "traffic_source": (("rank 0's slab saved to a snapshot and stepped ALONE (no neighbours: its boundary layers see no "
                                             "ghosts) by a child of bench.py; ") + traffic_note) if traffic else traffic_note,
                         "traffic_over_algorithmic": (traffic["force_fused"] / (84.0 * n_own)) if traffic else None,
                         "density_traffic": traffic.get("density") if traffic else None,
                         "algorithmic_bytes_per_particle": 84, "avg_launch_ms": t_force * 1e3,
                         "particles_rank0": int(n_own)},
            "cpu_baseline": cpu_base,  # rank 0's host, the 262,144-particle probe (None with --no-cpu)
            "gpu_over_cpu": (total * args.steps / wall / cpu_base["value"]) if cpu_base else None,
            "protocol": {"groups_per_step": sim.stats.get("protocol", 3), "one_message_steps": sim.stats.get("one_message_steps", 0),
                         "one_message_rests": sim.stats.get("one_message_rests", 0)},
            # ---- where the time of an N-rank step goes (DESIGN.md section 6 names the row each field confirms or refutes) ----
            "rccl": diag["rccl"],                       # None unless --transport rccl: {world_seen, devices, ping_us}
            "ping_us": diag["ping_us"],                 # preflight: one exchange-shaped group at the step's three message sizes
            "exchange_us": diag["exchange_us"],         # event pairs around each group, probe pass of `probe_steps` steps
            "host_wait_us": diag["host_wait_us"],       # the step's one host wait, timed window
            "host_step_us": diag["host_step_us"],       # host time in front of / behind the wait, whole call
            "probe_steps": probe,
            "early_force": sim.early_force,             # the innermost layers' force pass in front of the wait: on when a group costs > 30 us
            "phases_ms": diag["ranks"],                 # EVERY rank: phases, host timing, groups, pings
            "phases_ms_max_over_ranks": diag["phases_ms_max_over_ranks"],
            "phases_ms_rank0": phases_ms, "slab_stats_rank0": sim.stats, "owned_sum": int(counts[0]),
            "owned_max": int(n_max), "imbalance": float(n_max) * world / max(total, 1),
            "movers_per_step": movers_win, "sort_skips": int(counts[2]), "flowing": flow_ok,
        }
    sim.close()
    return out, (strict_flow and out is not None and not out["flowing"])


def bench_periodic(args):
    """`bench.py --force-slab --periodic-z [--lattice nx,ny,nz]`: ONE slab between its own periodic images (the loop
    transport, sph_loop_transport_create): the stand-in for a MIDDLE rank of an N-GPU run that a one-GPU box allows.  The
    slab does everything a rank between two neighbours does -- migrants, halo A and halo B at their real sizes, ghost unpack,
    the boundary layers' launches, the host wait -- with the device to itself; every message is held back on the comm stream by
    --link-latency-us + bytes / --link-gbs (a PARAMETER: 153 GB/s is one xGMI link's figure, not a measurement).  Default
    lattice 256,256,32: an eighth of config 3, the metric's 8-GPU point."""
    import ctypes as C

    import torch  # noqa: F401
    L = capi.load()
    cfg = ic.CONFIGS[getattr(args, "workload", "C3") or "C3"]
    nx, ny, nz = (int(v) for v in (args.lattice or "256,256,32").split(","))
    assert nz % 2 == 0 and nz >= 8, "whole cell layers, at least four of them"
    layers = nz // 2                                   # two lattice planes per cell layer (spacing 1/32, cells 1/16)
    z_lo, z_hi = layers, 2 * layers                    # away from the box's z faces: ghost layers on both sides
    box, grid = cfg["box"], cfg["grid"]
    assert z_hi + 1 < grid[2]
    n = nx * ny * nz
    edge = float(box[2]) / grid[2]
    params = capi.default_params(box, grid)
    per_layer = 2 * nx * ny
    protocol = int(getattr(args, "protocol", 3) or 3)
    ctx = capi.Context(int(1.5 * n) + 4096, params=params, device=0, slab=(z_lo, z_hi),
                       ghost_capacity=(5 if protocol == 1 else 3) * per_layer + 1024, ghost_layers=2 if protocol == 1 else 1)
    ctx.set_precision(getattr(args, "precision", "f32") == "mixed")
    ctx.reset_lattice((nx, ny, 2 * nz), jitter=True, jitter_dims=box, start=n, count=n)      # the planes of layers [z_lo, z_hi)
    tr = C.POINTER(capi.Transport)()
    capi._check(L.sph_loop_transport_create(C.byref(tr), layers * edge, float(args.link_gbs), float(args.link_latency_us)))
    h = C.c_void_p()
    capi._check(L.sph_slab_create(C.byref(h), ctx.h, 1, 3, tr, 0))
    if protocol == 1:
        capi._check(L.sph_slab_set_protocol(h, 1))
    ef_arg = getattr(args, "early_force", "auto")
    halo_us = float(args.link_latency_us) + (per_layer * 32 / (float(args.link_gbs) * 1e3) if float(args.link_gbs) > 0 else 0.0)
    ef_on = ef_arg == "on" or (ef_arg == "auto" and NativeSlabSimulation.early_force_rule(float(args.link_latency_us), halo_us, n)[0])
    capi._check(L.sph_slab_set_early_force(h, 1 if ef_on else 0))
    dt = float(ic.DEFAULT_DT)
    step = lambda k: capi._check(L.sph_slab_step(h, dt, int(k)))          # noqa: E731
    sync = lambda: capi._check(L.sph_slab_sync(h))                        # noqa: E731

    timing = lambda: slab_timing_dict(h)                                  # noqa: E731

    runup = 6000 if args.runup is None else args.runup
    tail = min(1000, runup)
    t0 = time.perf_counter()
    step(runup - tail); sync()
    q0, t_tail = ctx.sort_stats(), time.perf_counter()
    step(tail); sync()
    tail_wall = time.perf_counter() - t_tail
    q1 = ctx.sort_stats()
    print(f"[bench] periodic slab: run-up {runup} steps in {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    step(args.warmup); sync()
    capi._check(L.sph_slab_timing_reset(h))
    s0 = ctx.sort_stats()
    t0 = time.perf_counter()
    step(args.steps); sync()
    wall = time.perf_counter() - t0
    s1 = ctx.sort_stats()
    host_t = timing()
    ctx.timing(True); ctx.timing_reset()
    capi._check(L.sph_slab_timing_reset(h)); capi._check(L.sph_slab_timing_enable(h, 1))
    probe = max(2, min(args.steps, 50))
    step(probe); sync()
    ph, _ = ctx.timing_get()
    ctx.timing(False)
    probe_t = timing()
    capi._check(L.sph_slab_timing_enable(h, 0))
    stats = (C.c_uint64 * 8)()
    capi._check(L.sph_slab_counters(h, stats))
    owned = ctx.n
    st = ctx.download(index_base=n, count=n, want=("density", "vel"))
    finite = bool(np.isfinite(st["density"]).all() and np.isfinite(st["vel"]).all())
    out = {
        "metric": "particle-steps/sec (ONE slab between its periodic images: a middle rank's step, not a whole job)",
        "value": n * args.steps / wall, "unit": "particle-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "ms_per_step_sustained": tail_wall / max(tail, 1) * 1e3,
        "higher_is_better": True, "scaling": "n/a", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"dam-break slice {nx}x{ny}x{nz} = {n} particles on cell layers [{z_lo}, {z_hi}) of the {grid[2]}-layer grid, "
                               f"periodic images along z through the loop transport (shift {layers * edge}); every message held back by "
                               f"{args.link_latency_us} us + bytes / {args.link_gbs} GB/s; timed after {runup} run-up steps",
                   "particles": n, "link_gbs": float(args.link_gbs), "link_latency_us": float(args.link_latency_us), "runup_steps": runup,
                   "movers_per_step": (s1["movers_total"] - s0["movers_total"]) / max(args.steps, 1),
                   "movers_per_step_runup_tail": (q1["movers_total"] - q0["movers_total"]) / max(tail, 1)},
        "phases_ms": {k: v / probe for k, v in ph.items()}, "probe_steps": probe,
        "host_wait_us": host_t["host_wait_us"], "waits_ready_frac": host_t["waits_ready"] / max(host_t["steps"], 1),
        "host_step_us": {k: host_t[k] for k in ("host_pre_us", "host_post_us", "host_step_us")},
        "exchange_us": {g: probe_t["exchange_us_" + g] for g in _GROUPS},
        "slab_counters": dict(zip(("steps", "migrants", "resorts", "ghosts", "host_waits", "in_place_merges", "far_steps", "rest_messages"),
                                  (int(v) for v in stats))),
        "owned": int(owned), "finite": finite, "roofline": None, "cpu_baseline": None,
    }
    pr = (C.c_uint64 * 3)()
    capi._check(L.sph_slab_protocol(h, pr))
    out["protocol"] = {"groups_per_step": int(pr[0]), "one_message_steps": int(pr[1]), "one_message_rests": int(pr[2]),
                       "exchanges": int(L.sph_slab_exchanges(h))}
    ef = (C.c_uint64 * 2)()
    capi._check(L.sph_slab_early_force_stats(h, ef))
    out["early_force"] = {"on": bool(ef_on), "launched": int(ef[0]), "used": int(ef[1])}
    L.sph_slab_destroy(h)
    L.sph_loop_transport_destroy(tr)
    ctx.close()
    print(json.dumps(out), flush=True)
    if owned != n or not finite:
        sys.exit(f"bench: the periodic slab holds {owned} of {n} particles (finite: {finite})")


def bench_main(args):
    """bench.py --gpus N.  Under torch.distributed.run: one process per rank (RCCL, or gloo with --transport host).
    `--one-gpu` WITHOUT a launcher: the N ranks run as N threads of this one process on device 0 (a GPU box admits at
    most 6 processes on its card, so an 8-rank rehearsal cannot be 8 processes)."""
    import torch
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if getattr(args, "periodic_z", False):
        return bench_periodic(args)
    transport = getattr(args, "transport", "rccl")
    launched = "RANK" in os.environ
    if getattr(args, "one_gpu", False) and not launched and args.gpus > 1:
        if transport == "rccl":
            sys.exit("bench: --one-gpu needs --transport host or local (RCCL refuses two ranks on one device)")
        args.ranks_as = "threads of one process"
        torch.cuda.set_device(0)
        hub = LocalComm.Hub(args.gpus)
        # a rehearsal: a rank that fails must not leave the others in 120 s waits (the library's defaults)
        os.environ.setdefault("SPH_SLAB_TIMEOUT_S", "20")
        hub_tr = capi.LocalHub(args.gpus, timeout_s=20) if transport == "local" else None
        res, errors = [None] * args.gpus, []

        def rank_main(r):
            try:
                comm = LocalComm(hub, r)
                comm.local_hub = hub_tr
                res[r] = bench_rank(comm, 0, args, transport)
            except BaseException as e:     # noqa: BLE001
                import traceback
                traceback.print_exc()
                errors.append(e)
                hub.bar.abort()

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(args.gpus)]
        try:
            for t in threads: t.start()
            for t in threads: t.join()
        finally:
            if hub_tr is not None:
                hub_tr.close()
        if errors:
            sys.exit(f"bench: a rank failed: {errors[0]!r}")
        out = res[0][0]
        bad = any(r is not None and r[1] for r in res)          # any rank may have seen a non-flowing window
        print(json.dumps(out), flush=True)
        if bad:
            sys.exit("bench: the timed window was not a flowing state (sort skipped or too few particles changed cell)")
        return
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")     # only a direct single-rank run gets here without a launcher
    if getattr(args, "one_gpu", False):               # rehearsal on a one-GPU box: every rank on device 0, host-staged
        if transport != "host":
            sys.exit("bench: --one-gpu under a launcher needs --transport host (processes cannot share device pointers)")
        local = 0
    if transport == "local":
        sys.exit("bench: --transport local moves data between the streams of ONE process: use it with --one-gpu and no launcher")
    torch.cuda.set_device(local)
    # torch.distributed is the launcher's plumbing: rendezvous, the RCCL id, barriers and the max over ranks.  The
    # data path (migrants, halos) is the library's own RCCL communicator on its comm stream -- or, with
    # --transport host, host-staged messages over the process group (several ranks sharing one GPU: rehearsals).
    if transport == "rccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        comm = TorchDistComm(torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = TorchDistComm(torch.device("cpu"))
    args.ranks_as = "processes"
    out, bad = bench_rank(comm, local, args, transport)
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()
    if bad:
        sys.exit("bench: the timed window was not a flowing state (sort skipped or too few particles changed cell)")
