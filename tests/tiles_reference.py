"""TEST INFRASTRUCTURE: the Python reference implementation of the spatial-tile driver.

The product's tile step loop lives inside libsphx (`sphx_multi_*`, yasph2d_amd/csrc/sphx_tiles.cpp); this module restates the same
loop above the sub-step C ABI (`sphx_sub_*` / `sphx_tile_*`) so that the tests can check the C++ loop against it bit for bit
(tests/test_gpu_multi.py), run it over an oracle-backed backend on CPU (tests/test_tiles_cpu.py, tests/tile_oracle_backend.py)
and over a single GPU at the BASELINE sizes (tests/test_gpu_tiles_full.py).  Nothing under yasph2d_amd/ imports it.

The reference (yasph2d) has no distributed path.  The domain is cut at cell boundaries into rectangles — strips along one axis
(`StripLayout`, the 8-GPU layout) or columns that are cut again across (`GridLayout`, 2x2 on 4 GPUs); rank r OWNS the particles
whose cell lies in its rectangle and additionally holds GHOST copies of its neighbours' particles within `halo` cells of it.
One process per GPU; the only per-step collectives are

  * one halo exchange (32-byte particle records, send/recv with the <= 8 spatial neighbours; 2 for strips) between advect and
    re-grid:
    it carries migration (particles that crossed a cut) and rebuilds the ghost set from scratch, and
  * three tiny all-reduces: max |v + a dt|^2 (CFL, dfsph.rs:474-479) and the residual sum + owned count of every solver
    iteration (dfsph.rs:221, :377).

Ghost values are NOT exchanged per sub-step.  Every neighbour traversal makes the outermost still-valid ring of ghost cells
invalid (a ghost's own neighbours are missing beyond the halo), so with a halo of H cells the driver has a budget of H rings
between two exchanges; it tracks the budget (`_valid`) and inserts an extra exchange + re-grid only when a long solver loop
would exhaust it.  The redundant work is H cells per cut (a few % of a tile), the latency of 6 exchanges per step is saved.

The driver is backend-agnostic: `GpuTileBackend` (libsphx, the product) or the oracle-based backend of the CPU tests
(tests/tile_oracle_backend.py) implement the same sub-step interface, and `TorchComm` runs over RCCL ("nccl") on GPUs and
over gloo in the CPU tests.
"""
import contextlib
import ctypes as C
import threading

import numpy as np

from yasph2d_amd import _lib
from yasph2d_amd._lib import SphxError

HALO_RECORD_BYTES = 32
HALO_DTYPE = np.dtype([("pv", np.float32, 4), ("id", np.uint32), ("kappa", np.float32), ("stiff", np.float32), ("pad", np.uint32)])
OWNED_BIT = np.uint32(0x80000000)


def cell_coord(pos, axis, grid_min=-100.0, cell_inv=None, h=0.02):
    """GridProperties::position_to_mortoncellpos (neighborhood_search.rs:52-58) along one axis, same f32 ops as the device."""
    ci = np.float32(cell_inv) if cell_inv is not None else np.float32(1.0) / np.float32(h)
    v = (pos[:, axis].astype(np.float32) - np.float32(grid_min)) * ci
    return np.clip(np.nan_to_num(v, nan=0.0), 0.0, 65535.0).astype(np.uint32)


def quantile_cuts(coords, world):
    """Cut positions (cell indices) at particle-count quantiles; cuts[0] = 0, cuts[world] = 65536."""
    s = np.sort(coords)
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(s[(len(s) * r) // world]))
    cuts.append(65536)
    for r in range(1, world + 1):  # strictly increasing
        cuts[r] = max(cuts[r], cuts[r - 1] + 1)
    return cuts


# ------------------------------------------------------------------------------------------------------------ communicators
def rebalance_cuts(cuts, counts, halo, columns, max_shift, threshold=1.05):
    """Diffusive re-partition ("re-partition when max/mean load > ~1.1"): every interior cut moves towards the
    heavier of its two tiles by half their difference expressed in cell columns (`columns` = mean particles per column), at
    most `max_shift` cells (<= halo: the band that changes owner is already present as ghosts on the other side), and never
    below two halo widths per tile.  Pure function of rank-identical inputs, so every rank computes the same cuts."""
    W = len(counts)
    mean = sum(counts) / max(W, 1)
    if W < 2 or mean <= 0 or max(counts) <= threshold * mean:
        return list(cuts)
    new = list(cuts)
    for r in range(1, W):
        d = int(round((counts[r] - counts[r - 1]) * 0.5 / max(columns, 1.0)))
        new[r] = cuts[r] + max(-max_shift, min(max_shift, d))
    min_w = 2 * halo + 2
    for r in range(1, W):          # left to right: keep every tile wide enough
        lo_lim = new[r - 1] + min_w if r > 1 else new[r]
        new[r] = max(new[r], lo_lim)
    for r in range(W - 1, 0, -1):  # and right to left
        hi_lim = new[r + 1] - min_w if r < W - 1 else new[r]
        new[r] = min(new[r], hi_lim)
    for r in range(1, W):          # a cut that the width rule pushed further than max_shift stays where it was
        if abs(new[r] - cuts[r]) > max_shift:
            return list(cuts)
    return new


def grow(rect, halo):
    x0, x1, y0, y1 = rect
    return (max(0, x0 - halo), min(65536, x1 + halo), max(0, y0 - halo), min(65536, y1 + halo))


def rects_touch(a, b, halo):
    """True if rectangle b grown by `halo` cells overlaps a (the relation is symmetric)."""
    gx0, gx1, gy0, gy1 = b[0] - halo, b[1] + halo, b[2] - halo, b[3] + halo
    return a[0] < gx1 and gx0 < a[1] and a[2] < gy1 and gy0 < a[3]


def in_rect(cx, cy, rect, halo=0):
    x0, x1, y0, y1 = rect
    return (cx + halo >= x0) & (cx < x1 + halo) & (cy + halo >= y0) & (cy < y1 + halo)


class StripLayout:
    """`world` strips along `axis`; rank r owns [cuts[r], cuts[r+1]) x everything."""

    def __init__(self, axis, cuts):
        self.axis, self.cuts = int(axis), list(cuts)
        self.world = len(self.cuts) - 1

    def rects(self):
        full = (0, 65536)
        return [((self.cuts[r], self.cuts[r + 1]) + full) if self.axis == 0 else (full + (self.cuts[r], self.cuts[r + 1])) for r in range(self.world)]

    def state(self):
        return list(self.cuts)

    def rebalance(self, counts, halo, columns, max_shift):
        new = rebalance_cuts(self.cuts, counts, halo, columns[self.axis], max_shift)
        changed = new != self.cuts
        self.cuts = new
        return changed


class GridLayout:
    """nx columns cut at xcuts, each column cut again at its own ycuts[ix] ("x-cut then per-column y-cut");
    rank = ix * ny + iy."""

    def __init__(self, xcuts, ycuts):
        self.xcuts, self.ycuts = list(xcuts), [list(c) for c in ycuts]
        self.nx, self.ny = len(self.xcuts) - 1, len(self.ycuts[0]) - 1
        self.world = self.nx * self.ny

    @staticmethod
    def quantile(pos, nx, ny, grid_min=-100.0, h=0.02):
        cx, cy = cell_coord(pos, 0, grid_min, h=h), cell_coord(pos, 1, grid_min, h=h)
        xcuts = quantile_cuts(cx, nx)
        ycuts = []
        for ix in range(nx):
            m = (cx >= xcuts[ix]) & (cx < xcuts[ix + 1])
            ycuts.append(quantile_cuts(cy[m], ny) if m.any() else quantile_cuts(cy, ny))
        return GridLayout(xcuts, ycuts)

    def rects(self):
        return [(self.xcuts[ix], self.xcuts[ix + 1], self.ycuts[ix][iy], self.ycuts[ix][iy + 1]) for ix in range(self.nx) for iy in range(self.ny)]

    def state(self):
        return [list(self.xcuts)] + [list(c) for c in self.ycuts]

    def rebalance(self, counts, halo, columns, max_shift):
        before = self.state()
        col = [sum(counts[ix * self.ny:(ix + 1) * self.ny]) for ix in range(self.nx)]
        self.xcuts = rebalance_cuts(self.xcuts, col, halo, columns[0], max_shift)
        for ix in range(self.nx):
            # a column holds 1/nx of the particles: its rows are that much thinner
            self.ycuts[ix] = rebalance_cuts(self.ycuts[ix], counts[ix * self.ny:(ix + 1) * self.ny], halo, columns[1] / self.nx, max_shift)
        return self.state() != before


class TorchComm:
    """torch.distributed: backend "nccl" IS RCCL on ROCm (device tensors over xGMI); "gloo" in the CPU tests."""

    def __init__(self, dist, device):
        self.dist, self.device = dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def exchange(self, peers, sends, recvs, stream_ordered=False):
        """Send sends[k] to rank peers[k] and receive recvs[k] from it, all peers at once.  stream_ordered: the caller's kernels
        run on torch's CURRENT stream (GpuTileBackend.stream_context), so the collective library's stream semantics order
        pack -> send/recv -> unpack and no host synchronisation is needed."""
        import torch
        import torch.distributed as dist

        if not peers:
            return
        # gloo cannot send device tensors: stage through the host (functional tests on a single-GPU box); RCCL sends the
        # device buffers as they are.
        cuda = sends[0].is_cuda
        stage = dist.get_backend() == "gloo" and cuda
        ss = [t.cpu() for t in sends] if stage else sends
        rs = [torch.empty_like(t) for t in ss] if stage else recvs
        ops = []
        for k, peer in enumerate(peers):
            ops += [dist.P2POp(dist.isend, ss[k], peer), dist.P2POp(dist.irecv, rs[k], peer)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if stage:
            for k in range(len(peers)):
                recvs[k].copy_(rs[k])
        if cuda and not stream_ordered:
            torch.cuda.synchronize(sends[0].device)

    def _red_device(self):
        import torch

        return torch.device("cpu") if self.dist.get_backend() == "gloo" else self.device

    def allreduce_max(self, x):
        import torch

        t = torch.tensor([x], dtype=torch.float64, device=self._red_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allreduce_sum(self, xs):
        import torch

        t = torch.tensor(list(xs), dtype=torch.float64, device=self._red_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]


class ShmComm(TorchComm):
    """One node, one process per GPU: halo records over torch.distributed (RCCL), the per-step scalars over a POSIX
    shared-memory all-reduce in libsphx (they already sit in host memory on every rank; ~1 us instead of a device round trip)."""

    def __init__(self, dist, device, name):
        super().__init__(dist, device)
        self.L = _lib.lib()
        # collective: rank 0 replaces any stale segment of that name, the call returns when every rank has joined the new one
        self.h = self.L.sphx_shm_open(str(name).encode(), self.rank, self.world)
        if not self.h:
            raise RuntimeError("sphx_shm_open failed")
        self._in = (C.c_double * 8)()
        self._out = (C.c_double * 8)()

    def _reduce(self, xs, op):
        n = len(xs)
        for k, v in enumerate(xs):
            self._in[k] = v
        rc = self.L.sphx_shm_allreduce(self.h, self._in, n, op, self._out)
        if rc:
            raise SphxError(rc, "shared-memory all-reduce timed out")
        return [self._out[k] for k in range(n)]

    def allreduce_max(self, x):
        return self._reduce([x], 1)[0]

    def allreduce_sum(self, xs):
        return self._reduce(list(xs), 0)

    def close(self):
        if self.h:
            self.L.sphx_shm_close(self.h)
            self.h = None


class ThreadComm:
    """In-process communicator: `world` tiles run as threads of one process (single-GPU tests of the tile path)."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [dict() for _ in range(world)]

    def __init__(self, shared, rank):
        self.sh, self.rank, self.world = shared, rank, shared.world

    def exchange(self, peers, sends, recvs, stream_ordered=False):
        cuda = bool(sends) and sends[0].is_cuda
        if cuda:
            import torch

            torch.cuda.synchronize()  # the other thread reads these buffers from its own stream
        self.sh.slots[self.rank]["send"] = dict(zip(peers, sends))
        self.sh.barrier.wait()
        for k, peer in enumerate(peers):
            recvs[k].copy_(self.sh.slots[peer]["send"][self.rank])
        if cuda:
            torch.cuda.synchronize()
        self.sh.barrier.wait()

    def _gather(self, x):
        self.sh.slots[self.rank]["v"] = x
        self.sh.barrier.wait()
        vals = [self.sh.slots[r]["v"] for r in range(self.world)]
        self.sh.barrier.wait()
        return vals

    def allreduce_max(self, x):
        return max(self._gather(x))

    def allreduce_sum(self, xs):
        vals = self._gather(list(xs))
        return [sum(v[k] for v in vals) for k in range(len(xs))]


# ---------------------------------------------------------------------------------------------------------------- GPU backend
class GpuTileBackend:
    """The product path: one libsphx context per tile; halo buffers are torch device tensors handed to RCCL as they are."""

    def __init__(self, ctx, device=None, own_stream=True):
        import torch

        self.ctx = ctx
        self.L = ctx.L
        self.torch = torch
        self.device = device if device is not None else torch.device("cuda", ctx.params.device)
        # One torch stream carries the context's kernels AND the communication library's view of "current stream": packing,
        # RCCL send/recv and unpacking are ordered by the stream, the host never blocks in an exchange.
        self.stream = None
        if own_stream:
            self.stream = torch.cuda.Stream(self.device)
            self._chk(self.L.sphx_set_stream(ctx.h, C.c_void_p(self.stream.cuda_stream)))

    def stream_context(self):
        import contextlib

        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _chk(self, rc):
        if rc:
            raise SphxError(rc, self.L.sphx_last_error(self.ctx.h).decode())

    def make_buffers(self, cap, count):
        n = (1 + cap) * HALO_RECORD_BYTES
        with self.stream_context():
            bufs = [self.torch.zeros(n, dtype=self.torch.uint8, device=self.device) for _ in range(count)]
        self.torch.cuda.synchronize(self.device)
        return bufs

    def set_boundary(self, xy):
        self.ctx.set_boundary(xy)

    def configure(self, own, halo, peer_rects):
        rect = C.c_uint32 * 4
        peers = (rect * max(1, len(peer_rects)))(*[rect(*r) for r in peer_rects])
        self._chk(self.L.sphx_tile_configure_rect(self.ctx.h, rect(*own), halo, peers, len(peer_rects)))

    def reserve(self, capacity):
        self._chk(self.L.sphx_reserve(self.ctx.h, capacity))

    def upload(self, pos, vel, ids):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 2)
        ids = np.ascontiguousarray(ids, np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(self.L.sphx_tile_upload(self.ctx.h, p(pos), p(vel), p(ids), len(pos)))

    def pack(self, sends, cap):
        ptrs = (C.c_void_p * max(1, len(sends)))(*[t.data_ptr() for t in sends])
        self._chk(self.L.sphx_tile_pack_n(self.ctx.h, ptrs, len(sends), cap))

    def apply(self, recvs, cap):
        ptrs = (C.c_void_p * max(1, len(recvs)))(*[t.data_ptr() for t in recvs])
        self._chk(self.L.sphx_tile_apply_n(self.ctx.h, ptrs, len(recvs), cap))

    def regrid(self):
        n = C.c_uint32()
        self._chk(self.L.sphx_sub_regrid(self.ctx.h, C.byref(n)))
        return n.value

    def nonpressure(self, dt_prev):
        v = C.c_float()
        self._chk(self.L.sphx_sub_nonpressure(self.ctx.h, dt_prev, C.byref(v)))
        return v.value

    def predict(self, dt):
        self._chk(self.L.sphx_sub_predict(self.ctx.h, dt))

    def warmstart(self, divergence, dt):
        self._chk(self.L.sphx_sub_warmstart(self.ctx.h, int(divergence), dt))

    def iteration(self, divergence, dt, first):
        s, n = C.c_double(), C.c_uint64()
        self._chk(self.L.sphx_sub_iteration(self.ctx.h, int(divergence), dt, int(first), C.byref(s), C.byref(n)))
        return s.value, n.value

    def advect(self, dt):
        self._chk(self.L.sphx_sub_advect(self.ctx.h, dt))

    def synchronize(self):
        self.ctx.synchronize()

    def download(self):
        d = self.ctx.download()
        ss = self.ctx.download_solver_state()
        ids = d["ids"]
        return dict(pos=d["pos"], vel=d["vel"], density=d["density"], ids=ids & np.uint32(0x7FFFFFFF), owned=(ids >> np.uint32(31)) != 0,
                    kappa=ss["kappa"], stiffness=ss["stiffness"], alpha=ss["alpha"])


# -------------------------------------------------------------------------------------------------------------------- driver
class TiledDFSPH:
    """Solver::simulation_step (dfsph.rs:414-525) over spatial tiles.  All ranks call the same methods in lockstep.

    layout: StripLayout / GridLayout (identical on every rank); the historical form TiledDFSPH(backend, comm, axis, cuts, ...)
    builds a StripLayout."""

    def __init__(self, backend, comm, layout, cuts=None, halo=16, cap_records=None, max_avg_density_error=np.float32(0.01) / np.float32(100.0),
                 max_density_iterations=200, max_divergence_error=np.float32(0.1) / np.float32(100.0), max_divergence_iterations=400,
                 fixed_iterations=(0, 0), fluid_density=100.0, particle_radius=0.005, h=0.02, grid_min=-100.0, rebalance_every=0,
                 adaptive_halo=False, min_halo=6):
        if cuts is not None:
            layout = StripLayout(layout, cuts)
        self.b, self.comm, self.layout = backend, comm, layout
        # `halo` is the widest ghost band (buffers, tile widths and boundary clipping are sized for it); with adaptive_halo the band
        # in use follows the ring budget the last step actually needed
        self.halo_max = int(halo)
        self.adaptive_halo, self.min_halo = bool(adaptive_halo), min(int(min_halo), int(halo))
        self.halo_now = self.halo_max
        self._spent, self._last_div = [], (1, 0)  # ring consumption of the last intervals; (iterations, warm) of the last divergence loop
        self.rank, self.world = comm.rank, comm.world
        if layout.world != self.world:
            raise ValueError("layout and communicator disagree about the number of tiles")
        self.tol_d, self.max_d = np.float32(max_avg_density_error), int(max_density_iterations)
        self.tol_v, self.max_v = np.float32(max_divergence_error), int(max_divergence_iterations)
        self.fixed = fixed_iterations
        self.rho0 = np.float32(fluid_density)
        self.diam = np.float32(2.0) * np.float32(particle_radius)
        self.h, self.grid_min = h, grid_min
        self.num_density_iters, self.num_divergence_iters = 1, 0  # dfsph.rs:51,55
        self.cap = cap_records
        self.exchanges = 0
        self.rebalance_every, self.rebalances, self._steps = int(rebalance_every), 0, 0
        self.boundary_margin = 256  # extra cells of boundary particles kept on every side (cuts may drift that far before a re-clip)
        self._valid = self._kvalid = float("inf")
        self._bufs = {}

    @property
    def halo(self):
        return self.halo_max

    def _adapt_halo(self, spent):
        """Ghost band for the next exchange interval.  `spent` = rings the interval that is ending consumed (divergence loop of
        the previous step, non-pressure pass and density loop of this one).  The next band is the largest consumption of the last
        8 intervals plus one more iteration (2 rings), clamped to [min_halo, halo]; if a loop still runs longer, the ring budget
        triggers an extra exchange — slower, never wrong."""
        self._spent = (self._spent + [spent])[-8:]
        new = max(self.min_halo, min(self.halo_max, max(self._spent) + 2))
        if new != self.halo_now:
            self.halo_now = new
            self.b.configure(self.rect, self.halo_now, self.peer_rects)

    # historical accessors of the strip form (tests, bench)
    @property
    def axis(self):
        return getattr(self.layout, "axis", None)

    @property
    def cuts(self):
        return self.layout.state()

    # ---- geometry -----------------------------------------------------------------------------------------------------------
    def _place(self):
        """Own rectangle and the peers (tiles whose rectangle grown by the halo touches it), from the layout."""
        rects = self.layout.rects()
        self.rect = rects[self.rank]
        self.peers = [k for k in range(self.world) if k != self.rank and rects_touch(self.rect, rects[k], self.halo_max)]
        if len(self.peers) > 8:
            raise ValueError("a tile touches more than 8 others: tiles are too small for this halo")
        for k in [self.rank] + self.peers:  # interior extents must hold two halo widths (ghosts come from direct neighbours only)
            x0, x1, y0, y1 = rects[k]
            if (x0 > 0 and x1 < 65536 and x1 - x0 < 2 * self.halo_max) or (y0 > 0 and y1 < 65536 and y1 - y0 < 2 * self.halo_max):
                raise ValueError("tiles must be at least two halo widths wide")
        self.peer_rects = [rects[k] for k in self.peers]
        self.b.configure(self.rect, self.halo_now, self.peer_rects)

    def _buffers(self):
        need = [k for k in self.peers if k not in self._bufs]
        if need:
            new = self.b.make_buffers(self.cap, 2 * len(need))
            for j, k in enumerate(need):
                self._bufs[k] = (new[2 * j], new[2 * j + 1])
        return [self._bufs[k][0] for k in self.peers], [self._bufs[k][1] for k in self.peers]

    # ---- setup ------------------------------------------------------------------------------------------------------------
    def setup(self, pos, vel, ids, boundary):
        """All ranks pass the SAME global arrays (deterministic scene); each keeps its own cells."""
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        vel = np.zeros_like(pos) if vel is None else np.ascontiguousarray(vel, np.float32).reshape(-1, 2)
        ids = np.arange(len(pos), dtype=np.uint32) if ids is None else np.asarray(ids, np.uint32)
        cx, cy = cell_coord(pos, 0, self.grid_min, h=self.h), cell_coord(pos, 1, self.grid_min, h=self.h)
        self._place()
        mine = in_rect(cx, cy, self.rect)
        n_own = int(mine.sum())
        if self.cap is None:
            # particles this tile has to send to one peer: estimate from the global scene, with head-room for compression waves
            rects = self.layout.rects()
            near = 0
            for a in range(self.world):
                ma = in_rect(cx, cy, rects[a])
                for b in range(self.world):
                    if a != b and rects_touch(rects[a], rects[b], self.halo_max):
                        near = max(near, int((ma & in_rect(cx, cy, rects[b], self.halo_max)).sum()))
            self.cap = max(1024, int(near * 1.5) + 1024)
        self.b.reserve(int(n_own * 1.25) + 2 * max(2, len(self.peers)) * self.cap + 4096)
        span = lambda c: max(1, int(c.max()) - int(c.min()) + 1) if len(c) else 1  # noqa: E731
        self.columns = (len(pos) / span(cx), len(pos) / span(cy))  # mean particles per cell column / cell row
        self.n_owned_local = n_own
        self.boundary = None
        if boundary is not None and len(boundary):
            self.boundary = np.ascontiguousarray(boundary, np.float32).reshape(-1, 2)
            self.boundary_cells = (cell_coord(self.boundary, 0, self.grid_min, h=self.h), cell_coord(self.boundary, 1, self.grid_min, h=self.h))
            self._clip_boundary()
        self.b.upload(pos[mine], vel[mine], ids[mine])
        self.n_owned_global = len(pos)
        self.refresh()  # initial ghosts + the warm-up block (dfsph.rs:419-428): re-grid, densities, alpha

    def _clip_boundary(self):
        m = self.halo_max + 2 + (self.boundary_margin if self.rebalance_every else 0)
        self._clip_rect = self.rect
        keep = in_rect(self.boundary_cells[0], self.boundary_cells[1], self.rect, m)
        self.b.set_boundary(self.boundary[keep])

    # ---- load balance -------------------------------------------------------------------------------------------------------
    def _allgather(self, x):
        out = []
        for base in range(0, self.world, 8):  # the shared-memory reduction carries 8 doubles per call
            m = min(8, self.world - base)
            out += self.comm.allreduce_sum([float(x) if base + k == self.rank else 0.0 for k in range(m)])
        return out

    def rebalance(self):
        """Move the cuts towards equal owned counts; takes effect in the refresh() that follows (the pack/retire/apply rules are
        purely geometric, so particles of the band that changes owner travel as ordinary migrants)."""
        counts = self._allgather(self.n_owned_local)
        # a cut moves by at most halo/4 cells and never further than the narrowest ghost band in use: the band that changes owner is
        # then still inside the old owner's ghost band (it keeps those particles as ghosts, k_tile_pack)
        if not self.layout.rebalance(counts, self.halo_max, self.columns, max(1, min(self.halo_max // 4, self.min_halo))):
            return False
        self._place()
        if self.boundary is not None and max(abs(a - b) for a, b in zip(self.rect, self._clip_rect)) > self.boundary_margin // 2:
            self._clip_boundary()
        self.rebalances += 1
        return True

    # ---- halo ---------------------------------------------------------------------------------------------------------------
    def _cap_now(self):
        """Records per peer buffer for the band in use: the set-up estimate for the widest band, scaled (cuts may also have moved by
        up to halo/4 cells, hence the +4); every rank computes the same number, sender and receiver agree on the layout."""
        if self.halo_now >= self.halo_max:
            return self.cap
        return min(self.cap, max(1024, -(-self.cap * (self.halo_now + 4) // (self.halo_max + 4))))

    def refresh(self):
        """Halo exchange (migration + fresh ghosts) followed by the re-grid of the local set."""
        sends, recvs = self._buffers()
        cap = self._cap_now()
        nbytes = (1 + cap) * HALO_RECORD_BYTES
        sends, recvs = [t[:nbytes] for t in sends], [t[:nbytes] for t in recvs]  # only the part the band in use can fill travels
        self.b.pack(sends, cap)
        ordered = getattr(self.b, "stream", None) is not None  # kernels and communication share one stream: no host sync
        with (self.b.stream_context() if ordered else contextlib.nullcontext()):
            self.comm.exchange(self.peers, sends, recvs, stream_ordered=ordered)
        self.b.apply(recvs, cap)
        self.n_local = self.b.regrid()
        self.exchanges += 1
        full = float("inf") if self.world == 1 else float(self.halo_now)
        self._valid = self._kvalid = full          # rings (cells from the owned region) in which v* / kappa are exact
        self._avalid = full - 1                    # ... density and alpha (one traversal after the exchange)

    def _need(self, after):
        """Make sure the owned region stays exact after an operation that leaves `after` valid rings."""
        if after < 0:
            self.refresh()
            return True
        return False

    # ---- one step -----------------------------------------------------------------------------------------------------------
    def _loop(self, divergence, dt):
        prev = self.num_divergence_iters if divergence else self.num_density_iters
        fixed = self.fixed[1] if divergence else self.fixed[0]
        tol, cap = (self.tol_v, self.max_v) if divergence else (self.tol_d, self.max_d)
        warm = 0
        if prev > 1:  # dfsph.rs:199 / :354
            if self._need(min(self._valid, self._kvalid - 1)):
                pass
            self.b.warmstart(divergence, dt)
            self._valid = min(self._valid, self._kvalid - 1)
            warm = 1
        iters, avg = 0, np.float32(0)
        while True:
            self._need(min(self._valid - 2, self._avalid - 1))
            kvalid = min(self._valid - 1, self._avalid)
            s, n_owned = self.b.iteration(divergence, dt, iters == 0)
            self.n_owned_local = int(n_owned)
            self._valid = min(self._valid - 2, self._avalid - 1)
            self._kvalid = kvalid
            iters += 1
            S, N = self.comm.allreduce_sum([s, float(n_owned)])
            self.n_owned_global = int(N)
            avg = np.float32(np.float32(S) / np.float32(N))
            if divergence:
                avg = np.float32(avg / self.rho0)  # dfsph.rs:376-377
            if not np.isfinite(avg):
                raise SphxError(_lib.ERR_NONFINITE, "residual is not finite (dfsph.rs:223,378)")
            if fixed:
                if iters >= fixed:
                    break
                continue
            rel = avg if divergence else np.float32(avg / self.rho0)  # dfsph.rs:222
            if np.float32(rel * np.float32(dt)) < tol:               # dfsph.rs:226 / :381
                break
            if iters > cap:                                          # dfsph.rs:236 / :391
                break
        if divergence:
            self.num_divergence_iters = iters
        else:
            self.num_density_iters = iters
        return iters, float(avg), warm

    def step(self, timer):
        """One simulation_step; `timer` mirrors TimeManager (yasph2d_amd.TimeManager); identical on every rank."""
        from yasph2d_amd import duration_as_secs_f32

        dt_prev = timer.simulation_step()
        self._need(min(self._avalid, self._valid) - 1)
        vsq = self.b.nonpressure(dt_prev)                                   # dfsph.rs:436-477
        vmax = float(np.sqrt(np.float32(self.comm.allreduce_max(float(vsq)))))
        dt_ns = timer.update_simulation_step(self.diam, vmax)               # dfsph.rs:478-480
        dt = duration_as_secs_f32(dt_ns)
        self.b.predict(dt)                                                  # dfsph.rs:484-492
        self._valid = min(self._avalid, self._valid) - 1
        Id, avg_d, wd = self._loop(False, dt)                               # dfsph.rs:496
        self.b.advect(dt)                                                   # dfsph.rs:499-510 (ghosts move with their exact copies' v*)
        self._steps += 1
        if self.rebalance_every and self.world > 1 and self._steps % self.rebalance_every == 0:
            self.rebalance()
        if self.adaptive_halo and self.world > 1:
            # rings of the interval that ends here: divergence loop of the previous step, non-pressure pass, this density loop, and
            # the one-ring offset of density/alpha (computed one traversal after the exchange)
            self._adapt_halo(self._last_div[1] + 2 * self._last_div[0] + 1 + wd + 2 * Id + 1)
        self.refresh()                                                      # migration + ghosts, dfsph.rs:512-518
        Iv, avg_v, wv = self._loop(True, dt)                                # dfsph.rs:521
        self._last_div = (Iv, wv)
        return dict(density_iterations=Id, divergence_iterations=Iv, warmstart_density=wd, warmstart_divergence=wv, avg_density_error=avg_d,
                    avg_divergence=avg_v, dt_prev=dt_prev, dt=dt, vmax=vmax, dt_ns=dt_ns, n_local=self.n_local, n_global=self.n_owned_global)

    def download_owned(self):
        d = self.b.download()
        m = d["owned"]
        return {k: (v[m] if hasattr(v, "__len__") and len(v) == len(m) else v) for k, v in d.items()}
