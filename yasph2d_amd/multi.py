"""ctypes view of sphx_multi — the multi-GPU solver INSIDE libsphx (include/sphx.h, csrc/sphx_tiles.cpp).

`MultiSolver(devices=[...])` holds all tiles in this process (what a Rust host gets behind its one `Box<dyn Solver>`);
`MultiSolver.rank(...)` is ONE tile of a one-process-per-GPU run: with `dist=None` the library exchanges the halo records itself
(grouped ncclSend/ncclRecv over RCCL, shared-memory scalars), with a `torch.distributed` module the exchange is handed to it through
the sphx_comm_ops function table (the functional tests run that over gloo on a one-GPU box, where RCCL refuses two ranks per GPU).
The step loop is not here: it is C++ inside the library.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SphxError, SphxMultiInfo, SphxMultiOptions, SphxStepStats


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class TorchCommOps:
    """sphx_comm_ops over torch.distributed: halo records via batch_isend_irecv (nccl = RCCL: device buffers as they are; gloo:
    staged through the host), scalars via all_reduce (or the library's shared-memory reduction when `shm_name` is given)."""

    def __init__(self, dist, device, shm_name=None):
        import torch

        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.L = _lib.lib()
        self.shm = None
        if shm_name is not None:
            self.shm = self.L.sphx_shm_open(str(shm_name).encode(), self.rank, self.world)  # collective: returns when all ranks have joined
            if not self.shm:
                raise RuntimeError("sphx_shm_open failed")
        self._ex = _lib.COMM_EXCHANGE(self._exchange)
        self._ar = _lib.COMM_ALLREDUCE(self._allreduce)
        self._ab = _lib.COMM_ABORT(self._abort)
        self.ops = _lib.SphxCommOps(None, self.rank, self.world, self._ex, self._ar, self._ab)
        self.error = None

    def _tensor(self, ptr, nbytes):
        # a uint8 view of a device buffer libsphx owns (no copy): __cuda_array_interface__ v2
        holder = type("DevBuf", (), {"__cuda_array_interface__": {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}})()
        return self.torch.as_tensor(holder, device=self.device)

    def _exchange(self, user, peers, n_peers, d_send, d_recv, nbytes, hip_stream):
        try:
            torch, dist = self.torch, self.dist
            stream = torch.cuda.ExternalStream(int(hip_stream), device=self.device)
            with torch.cuda.stream(stream):
                sends = [self._tensor(d_send[k], nbytes) for k in range(n_peers)]
                recvs = [self._tensor(d_recv[k], nbytes) for k in range(n_peers)]
                stage = dist.get_backend() == "gloo"
                if stage:
                    stream.synchronize()
                ss = [t.cpu() for t in sends] if stage else sends
                rs = [torch.empty_like(t) for t in ss] if stage else recvs
                ops = []
                for k in range(n_peers):
                    ops += [dist.P2POp(dist.isend, ss[k], int(peers[k])), dist.P2POp(dist.irecv, rs[k], int(peers[k]))]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                if stage:
                    for k in range(n_peers):
                        recvs[k].copy_(rs[k])
            return 0
        except BaseException as e:  # noqa: BLE001 - must not unwind through the C frame
            self.error = e
            return _lib.ERR_HIP

    def _allreduce(self, user, pin, n, op, pout):
        try:
            if self.shm:
                return self.L.sphx_shm_allreduce(self.shm, C.cast(pin, C.c_void_p), n, op, C.cast(pout, C.c_void_p))
            torch, dist = self.torch, self.dist
            dev = torch.device("cpu") if dist.get_backend() == "gloo" else self.device
            t = torch.tensor([pin[k] for k in range(n)], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
            for k, v in enumerate(t.tolist()):
                pout[k] = v
            return 0
        except BaseException as e:  # noqa: BLE001
            self.error = e
            return _lib.ERR_HIP

    def _abort(self, user):
        if self.shm:  # (torch.distributed has no abort: its collectives run into their own time-out)
            self.L.sphx_shm_abort(self.shm)

    def close(self):
        if self.shm:
            self.L.sphx_shm_close(self.shm)
            self.shm = None


class MultiSolver:
    def __init__(self, params, devices=None, halo=16, fixed_halo=False, rebalance_every=16, layout=_lib.LAYOUT_AUTO, cap_records=0,
                 overlap_exchange=False, _rank=None):
        self.L = _lib.lib()
        self.params = params
        o = SphxMultiOptions()
        self.L.sphx_multi_default_options(C.byref(o))
        o.halo_cells, o.fixed_halo, o.rebalance_every, o.layout, o.cap_records = halo, int(fixed_halo), rebalance_every, layout, cap_records
        o.overlap_exchange = int(overlap_exchange)
        self.options = o
        h = C.c_void_p()
        if _rank is None:
            devs = (C.c_int * len(devices))(*devices)
            rc = self.L.sphx_multi_create(C.byref(params), devs, len(devices), C.byref(o), C.byref(h))
        else:
            device, comm, job, rank, world = _rank
            self._comm = comm  # keeps the callbacks alive
            rc = self.L.sphx_multi_create_rank(C.byref(params), device, C.byref(comm.ops) if comm is not None else None,
                                               str(job).encode() if job is not None else None, rank, world, C.byref(o), C.byref(h))
        if rc:
            raise SphxError(rc, self.L.sphx_multi_last_error(None).decode())
        self.h = h

    @classmethod
    def rank(cls, params, device, rank, world, comm=None, job=None, **kw):
        """One tile of a multi-process run.  comm: TorchCommOps, or None for the library's own RCCL + shared-memory transport."""
        return cls(params, _rank=(device, comm, job, rank, world), **kw)

    def close(self):
        if getattr(self, "h", None):
            self.L.sphx_multi_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            msg = self.L.sphx_multi_last_error(self.h).decode()
            comm = getattr(self, "_comm", None)
            if comm is not None and comm.error is not None:
                msg += f" (communicator: {comm.error!r})"
            raise SphxError(rc, msg)

    def set_strips(self, axis, cuts):
        c = np.ascontiguousarray(cuts, np.uint32)
        self._chk(self.L.sphx_multi_set_layout(self.h, axis, _p(c), len(c)))

    def set_grid(self, xcuts, ycuts):
        x = np.ascontiguousarray(xcuts, np.uint32)
        yc = np.ascontiguousarray(ycuts, np.uint32)
        self._chk(self.L.sphx_multi_set_grid_layout(self.h, len(x) - 1, yc.shape[1] - 1, _p(x), _p(yc)))

    def set_boundary(self, xy):
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        self._chk(self.L.sphx_multi_set_boundary(self.h, _p(xy), len(xy)))

    def upload(self, pos, vel=None, ids=None):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 2) if vel is not None else None
        ids = np.ascontiguousarray(ids, np.uint32) if ids is not None else None
        self._chk(self.L.sphx_multi_upload(self.h, _p(pos), _p(vel), _p(ids), len(pos)))

    def clear_cached(self):
        self._chk(self.L.sphx_multi_clear_cached(self.h))

    def step_begin(self, dt_prev):
        v = C.c_float()
        self._chk(self.L.sphx_multi_step_begin(self.h, dt_prev, C.byref(v)))
        return v.value

    def step_finish(self, dt):
        st = SphxStepStats()
        self._chk(self.L.sphx_multi_step_finish(self.h, dt, C.byref(st)))
        return st.as_dict()

    def step(self, timer, diameter=np.float32(0.01)):
        """Solver::simulation_step with the host mirror of the caller's TimeManager in the middle (dfsph.rs:478-480)."""
        st = SphxStepStats()
        self._chk(self.L.sphx_multi_simulation_step(self.h, timer.h, diameter, C.byref(st)))
        d = st.as_dict()
        d["dt_ns"] = timer.simulation_step_ns()
        return d

    def steps(self, timer, k, diameter=np.float32(0.01)):
        """k consecutive step() calls inside the library (the frame loop of main.rs:348-350); returns the k stats dicts."""
        st = (SphxStepStats * k)()
        done = C.c_uint32()
        self._chk(self.L.sphx_multi_simulation_steps(self.h, timer.h, diameter, k, st, C.byref(done)))
        return [x.as_dict() for x in st]

    def synchronize(self):
        self._chk(self.L.sphx_multi_synchronize(self.h))

    def download(self):
        n = C.c_uint64(self.L.sphx_multi_num_owned(self.h) + 1024)
        for _ in range(2):
            cap = n.value
            pos, vel = np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32)
            den, ids = np.zeros(cap, np.float32), np.zeros(cap, np.uint32)
            rc = self.L.sphx_multi_download(self.h, _p(pos), _p(vel), _p(den), _p(ids), C.byref(n))
            if rc != _lib.ERR_CAPACITY:
                break
        self._chk(rc)
        k = n.value
        return dict(pos=pos[:k], vel=vel[:k], density=den[:k], ids=ids[:k])

    def info(self):
        i = SphxMultiInfo()
        self._chk(self.L.sphx_multi_info(self.h, C.byref(i)))
        d = {k: getattr(i, k) for k, _ in i._fields_ if k not in ("reserved", "transport")}
        d["transport"] = i.transport.decode()
        return d

    def set_tiling_invariant(self, on=True):
        """sphx_set_tiling_invariant on every local tile: cell mates ordered by persistent id (the tiles' warm-start values always travel).
        A single context and the oracle in the same mode then compute the same run, bit for bit (a comparison mode)."""
        for k in range(self.info()["local_tiles"]):
            self.tile_context(k).set_tiling_invariant(on)

    def tile_context(self, k=0):
        """Borrowed SphxContext view of a local tile (inspection, profiling)."""
        from . import SphxContext

        ctx = SphxContext.__new__(SphxContext)
        ctx.L, ctx.params, ctx._owned = self.L, None, False
        ctx.h = C.c_void_p(self.L.sphx_multi_tile_ctx(self.h, k))
        return ctx
