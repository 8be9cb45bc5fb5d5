"""CPU backend of tests/tiles_reference.TiledDFSPH for the tests: the sub-steps are executed by the oracle, the halo
pack/apply logic is restated in numpy (same selection rules and the same record order as the HIP kernels k_tile_*)."""
import ctypes as C

import numpy as np
import torch

from oracle.oracle import Oracle
from tiles_reference import HALO_DTYPE, HALO_RECORD_BYTES, cell_coord, in_rect

OWNED = np.uint32(0x80000000)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleTileBackend:
    def __init__(self):
        self.o = Oracle()
        self.L = self.o.L

    def make_buffers(self, cap, count):
        return [torch.zeros((1 + cap) * HALO_RECORD_BYTES, dtype=torch.uint8) for _ in range(count)]

    def set_boundary(self, xy):
        self.o.set_boundary(xy)

    def configure(self, own, halo, peer_rects):
        self.rect, self.halo, self.peer_rects = tuple(own), halo, [tuple(r) for r in peer_rects]
        self.L.orc_tile_configure_rect(self.o.h, *self.rect)

    def reserve(self, capacity):
        pass

    def _set(self, pos, vel, ids, kappa, stiff):
        pos, vel = np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(vel, np.float32)
        ids = np.ascontiguousarray(ids, np.uint32)
        kappa, stiff = np.ascontiguousarray(kappa, np.float32), np.ascontiguousarray(stiff, np.float32)
        self.L.orc_tile_set_state(self.o.h, _p(pos), _p(vel), _p(ids), _p(kappa), _p(stiff), len(pos))

    def _state(self):
        n = self.o.n
        return self.o.positions(), self.o.velocities(), self.o.ids(), (self.o.kappa() if n else np.zeros(0, np.float32)), (
            self.o.stiffness() if n else np.zeros(0, np.float32))

    def _cells(self, pos):
        return cell_coord(pos, 0), cell_coord(pos, 1)

    def upload(self, pos, vel, ids):
        n = len(pos)
        self._set(pos, vel, np.asarray(ids, np.uint32) | OWNED, np.zeros(n, np.float32), np.zeros(n, np.float32))

    def pack(self, sends, cap):
        """k_tile_count/offsets/pack: for every peer the owned particles inside its rectangle grown by the halo, ascending index."""
        pos, vel, ids, kappa, stiff = self._state()
        owned = ((ids >> np.uint32(31)) != 0) & ~np.isnan(pos[:, 0]) if len(pos) else np.zeros(0, bool)
        cx, cy = self._cells(pos)
        for buf, rect in zip(sends, self.peer_rects):
            rec = buf.numpy().view(HALO_DTYPE)
            rec[:] = 0
            idx = np.nonzero(owned & in_rect(cx, cy, rect, self.halo))[0]
            assert len(idx) <= cap, "halo buffer too small"
            rec["id"][0] = len(idx)
            r = rec[1:1 + len(idx)]
            r["pv"][:, :2] = pos[idx]
            r["pv"][:, 2:] = vel[idx]
            r["id"] = ids[idx] & np.uint32(0x7FFFFFFF)
            r["kappa"] = kappa[idx]
            r["stiff"] = stiff[idx]

    def apply(self, recvs, cap):
        pos, vel, ids, kappa, stiff = self._state()
        owned = (ids >> np.uint32(31)) != 0
        cx, cy = self._cells(pos)
        own = owned & in_rect(cx, cy, self.rect)
        # owned particles that crossed a cut stay as ghosts while they are inside the ghost band (k_tile_pack's retire rule)
        keep = own | (owned & in_rect(cx, cy, self.rect, self.halo))
        ids = np.where(own, ids, ids & np.uint32(0x7FFFFFFF)).astype(np.uint32)
        parts = [(pos[keep], vel[keep], ids[keep], kappa[keep], stiff[keep])]
        for buf in recvs:
            rec = buf.numpy().view(HALO_DTYPE)
            cnt = int(rec["id"][0])
            r = rec[1:1 + cnt]
            p, v = r["pv"][:, :2].copy(), r["pv"][:, 2:].copy()
            ccx, ccy = self._cells(p)
            o = in_rect(ccx, ccy, self.rect)
            m = in_rect(ccx, ccy, self.rect, self.halo)
            parts.append((p[m], v[m], (r["id"] | np.where(o, OWNED, np.uint32(0)).astype(np.uint32))[m], r["kappa"][m], r["stiff"][m]))
        self._set(*[np.concatenate([q[k] for q in parts]) for k in range(5)])

    def regrid(self):
        self.L.orc_sub_regrid(self.o.h)
        return self.o.n

    def nonpressure(self, dt_prev):
        return float(self.L.orc_sub_nonpressure(self.o.h, dt_prev))

    def predict(self, dt):
        self.L.orc_sub_predict(self.o.h, dt)

    def warmstart(self, divergence, dt):
        self.L.orc_sub_warmstart(self.o.h, int(divergence), dt)

    def iteration(self, divergence, dt, first):
        s = float(self.L.orc_sub_iteration(self.o.h, int(divergence), dt, int(first)))
        pos, _, ids, _, _ = self._state()
        owned = ((ids >> np.uint32(31)) != 0)
        return s, int(owned.sum())

    def advect(self, dt):
        self.L.orc_sub_advect(self.o.h, dt)

    def synchronize(self):
        pass

    def download(self):
        pos, vel, ids, kappa, stiff = self._state()
        return dict(pos=pos, vel=vel, density=self.o.densities(), ids=ids & np.uint32(0x7FFFFFFF), owned=(ids >> np.uint32(31)) != 0, kappa=kappa,
                    stiffness=stiff, alpha=self.o.alpha())
