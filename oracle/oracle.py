"""ctypes loader for the CPU oracle (oracle/sph_oracle.cpp).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under yasph2d_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    """Compile liboracle.so / liboracle_omp.so with the committed Makefile (g++ only)."""
    need = force or not all(os.path.exists(os.path.join(_HERE, n)) for n in ("liboracle.so", "liboracle_omp.so"))
    if not need:
        src = os.path.getmtime(os.path.join(_HERE, "sph_oracle.cpp"))
        need = any(os.path.getmtime(os.path.join(_HERE, n)) < src for n in ("liboracle.so", "liboracle_omp.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-j2"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)


class OrcParams(C.Structure):
    _fields_ = [
        ("smoothing_factor", C.c_float),
        ("particle_density", C.c_float),
        ("fluid_density", C.c_float),
        ("gravity_x", C.c_float),
        ("gravity_y", C.c_float),
        ("grid_min_x", C.c_float),
        ("grid_min_y", C.c_float),
        ("search_radius", C.c_float),
    ]


class StepStats(C.Structure):
    _fields_ = [
        ("density_iterations", C.c_uint32),
        ("divergence_iterations", C.c_uint32),
        ("warmstart_density", C.c_uint32),
        ("warmstart_divergence", C.c_uint32),
        ("avg_density_error", C.c_float),
        ("avg_divergence", C.c_float),
        ("dt_prev", C.c_float),
        ("dt", C.c_float),
        ("vmax", C.c_float),
        ("neighbor_flags", C.c_uint32),
        ("pad", C.c_uint32),
        ("neighbor_entries", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


_libs = {}
# OrcPhase order of orc_phase_seconds (sph_oracle.cpp); "serial" = a loop the reference leaves serial
PHASES = ["cell_indices (serial)", "sort", "apply_sorting (serial)", "cell_array (serial)", "neighbor_lists", "update_densities", "compute_alpha_factors",
          "nonpressure_forces", "max_velocity+predict (serial)", "density_loop", "advect", "divergence_loop"]


def phase_seconds(L):
    """{phase: seconds} accumulated by this library instance since orc_phase_reset."""
    buf = (C.c_double * len(PHASES))()
    L.orc_phase_seconds(buf, len(PHASES))
    return {PHASES[k]: float(buf[k]) for k in range(len(PHASES))}



def lib(omp=False):
    key = "omp" if omp else "st"
    if key in _libs:
        return _libs[key]
    build()
    L = C.CDLL(os.path.join(_HERE, "liboracle_omp.so" if omp else "liboracle.so"))
    u16, u32, u64, f32, i32, vp = C.c_uint16, C.c_uint32, C.c_uint64, C.c_float, C.c_int, C.c_void_p
    sig = {
        "orc_morton_encode_lookup": (u32, [u16, u16]),
        "orc_morton_encode_bitfiddle": (u32, [u16, u16]),
        "orc_morton_decode_x": (u32, [u32]),
        "orc_morton_decode_y": (u32, [u32]),
        "orc_morton_is_in_rect": (i32, [u32, u32, u32]),
        "orc_morton_find_bigmin": (u32, [u32, u32, u32]),
        "orc_duration_from_secs_f32": (u64, [f32]),
        "orc_duration_as_secs_f32": (f32, [u64]),
        "orc_powi": (f32, [f32, i32]),
        "orc_kernel_evaluate": (f32, [i32, f32, f32, f32]),
        "orc_kernel_gradient": (None, [i32, f32, f32, f32, f32, f32, vp]),
        "orc_kernel_constants": (None, [i32, f32, vp]),
        "orc_set_threads": (None, [i32]),
        "orc_get_max_threads": (i32, []),
        "orc_set_all_parallel": (None, [i32]),
        "orc_get_proc_bind": (i32, []),
        "orc_get_num_places": (i32, []),
        "orc_create": (vp, [C.POINTER(OrcParams)]),
        "orc_destroy": (None, [vp]),
        "orc_get_properties": (None, [vp, vp]),
        "orc_timer_adaptive": (None, [vp, u64, u64, f32]),
        "orc_timer_fixed": (None, [vp, u64]),
        "orc_timer_target_frame": (None, [vp, u64]),
        "orc_timer_on_step_started": (None, [vp]),
        "orc_timer_step_ns": (u64, [vp]),
        "orc_timer_update": (u64, [vp, f32, f32]),
        "orc_set_boundary": (None, [vp, vp, u32]),
        "orc_set_particles": (None, [vp, vp, vp, u32]),
        "orc_num_particles": (u32, [vp]),
        "orc_num_boundary": (u32, [vp]),
        "orc_update_neighborhood": (None, [vp]),
        "orc_update_densities": (None, [vp, i32]),
        "orc_compute_alpha": (None, [vp]),
        "orc_dfsph_set_fixed_iterations": (None, [vp, u32, u32]),
        "orc_dfsph_set_warmstart_travel": (None, [vp, i32]),
        "orc_set_tiling_invariant": (None, [vp, i32]),
        "orc_phase_seconds": (i32, [C.POINTER(C.c_double), i32]),
        "orc_phase_reset": (None, []),
        "orc_dfsph_set_tolerances": (None, [vp, f32, u32, f32, u32]),
        "orc_dfsph_clear_cached": (None, [vp]),
        "orc_dfsph_step": (None, [vp, C.POINTER(StepStats)]),
        "orc_wcsph_clear_cached": (None, [vp]),
        "orc_wcsph_step": (None, [vp, C.POINTER(StepStats)]),
        "orc_get_positions": (None, [vp, vp]),
        "orc_get_velocities": (None, [vp, vp]),
        "orc_get_boundary": (None, [vp, vp]),
        "orc_get_densities": (None, [vp, vp]),
        "orc_get_ids": (None, [vp, vp]),
        "orc_get_boundary_ids": (None, [vp, vp]),
        "orc_get_alpha": (u32, [vp, vp]),
        "orc_get_kappa": (None, [vp, vp]),
        "orc_get_stiffness": (None, [vp, vp]),
        "orc_get_cells": (u32, [vp, i32, vp, vp]),
        "orc_get_neighbor_counts": (u64, [vp, vp]),
        "orc_get_neighbor_lists": (None, [vp, vp]),
        "orc_get_neighbor_flags": (u32, [vp]),
        "orc_tile_configure": (None, [vp, i32, u32, u32]),
        "orc_tile_configure_rect": (None, [vp, u32, u32, u32, u32]),
        "orc_tile_set_state": (None, [vp, vp, vp, vp, vp, vp, u32]),
        "orc_sub_regrid": (None, [vp]),
        "orc_sub_nonpressure": (f32, [vp, f32]),
        "orc_sub_predict": (None, [vp, f32]),
        "orc_sub_warmstart": (None, [vp, i32, f32]),
        "orc_sub_iteration": (C.c_double, [vp, i32, f32, i32]),
        "orc_sub_advect": (None, [vp, f32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _libs[key] = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


KERNEL_WENDLAND, KERNEL_POLY6, KERNEL_SPIKY = 0, 1, 2


class Oracle:
    """One reference-style simulation: FluidParticleWorld + DFSPHSolver/WCSPHSolver + TimeManager (CPU)."""

    def __init__(self, smoothing_factor=2.0, particle_density=10000.0, fluid_density=100.0, gravity=(0.0, -9.81),
                 grid_min=(-100.0, -100.0), search_radius=0.0, omp=False):
        self.L = lib(omp)
        p = OrcParams(smoothing_factor, particle_density, fluid_density, gravity[0], gravity[1], grid_min[0], grid_min[1],
                      search_radius)
        self.h = self.L.orc_create(C.byref(p))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    # --- setup
    def properties(self):
        out = np.zeros(4, np.float32)
        self.L.orc_get_properties(self.h, _ptr(out))
        return dict(smoothing_length=out[0], particle_mass=out[1], particle_radius=out[2], fluid_density=out[3])

    def set_boundary(self, xy):
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        self.L.orc_set_boundary(self.h, _ptr(xy), len(xy))

    def set_particles(self, pos, vel=None):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        if vel is not None:
            vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 2)
        self.L.orc_set_particles(self.h, _ptr(pos), _ptr(vel) if vel is not None else None, len(pos))

    def timer_adaptive(self, tmax_ns, tmin_ns, cfl):
        self.L.orc_timer_adaptive(self.h, tmax_ns, tmin_ns, cfl)

    def timer_target_frame(self, target_ns):
        self.L.orc_timer_target_frame(self.h, target_ns)

    def timer_on_step_started(self):
        self.L.orc_timer_on_step_started(self.h)

    def timer_fixed(self, step_ns):
        self.L.orc_timer_fixed(self.h, step_ns)

    def timer_step_ns(self):
        return self.L.orc_timer_step_ns(self.h)

    def set_fixed_iterations(self, nd, nv):
        self.L.orc_dfsph_set_fixed_iterations(self.h, nd, nv)

    def set_tiling_invariant(self, on=True):
        """NOT the reference's behaviour (sphx_set_tiling_invariant's twin): the particles of a cell are ordered by persistent id and the
        warm-start values travel with their particle — the two places where a run depends on how the domain is cut into tiles."""
        self.L.orc_set_tiling_invariant(self.h, int(bool(on)))

    def set_warmstart_travel(self, on=True):
        """NOT the reference's behaviour: warmstart_kappa / warmstart_stiffness are permuted with their particle on every re-sort
        (the reference leaves them slot-bound, dfsph.rs:512).  The mode multi-GPU tilings are compared with a single domain in."""
        self.L.orc_dfsph_set_warmstart_travel(self.h, int(bool(on)))

    # --- path pieces
    def update_neighborhood(self):
        self.L.orc_update_neighborhood(self.h)

    def update_densities(self, kind=KERNEL_WENDLAND):
        self.L.orc_update_densities(self.h, kind)

    def compute_alpha(self):
        self.L.orc_compute_alpha(self.h)

    def dfsph_step(self):
        st = StepStats()
        self.L.orc_dfsph_step(self.h, C.byref(st))
        return st.as_dict()

    def wcsph_step(self):
        st = StepStats()
        self.L.orc_wcsph_step(self.h, C.byref(st))
        return st.as_dict()

    def clear_cached(self):
        self.L.orc_dfsph_clear_cached(self.h)
        self.L.orc_wcsph_clear_cached(self.h)

    # --- getters
    @property
    def n(self):
        return self.L.orc_num_particles(self.h)

    @property
    def nb(self):
        return self.L.orc_num_boundary(self.h)

    def _get(self, fn, shape, dtype):
        out = np.zeros(shape, dtype)
        fn(self.h, _ptr(out))
        return out

    def positions(self):
        return self._get(self.L.orc_get_positions, (self.n, 2), np.float32)

    def velocities(self):
        return self._get(self.L.orc_get_velocities, (self.n, 2), np.float32)

    def boundary(self):
        return self._get(self.L.orc_get_boundary, (self.nb, 2), np.float32)

    def densities(self):
        return self._get(self.L.orc_get_densities, (self.n,), np.float32)

    def ids(self):
        return self._get(self.L.orc_get_ids, (self.n,), np.uint32)

    def boundary_ids(self):
        return self._get(self.L.orc_get_boundary_ids, (self.nb,), np.uint32)

    def alpha(self):
        m = self.L.orc_get_alpha(self.h, None)
        out = np.zeros(m, np.float32)
        self.L.orc_get_alpha(self.h, _ptr(out))
        return out

    def kappa(self):
        return self._get(self.L.orc_get_kappa, (self.L.orc_get_alpha(self.h, None),), np.float32)

    def stiffness(self):
        return self._get(self.L.orc_get_stiffness, (self.L.orc_get_alpha(self.h, None),), np.float32)

    def cells(self, static=False):
        m = self.L.orc_get_cells(self.h, int(static), None, None)
        first = np.zeros(m, np.uint32)
        cidx = np.zeros(m, np.uint32)
        self.L.orc_get_cells(self.h, int(static), _ptr(first), _ptr(cidx))
        return first, cidx

    def neighbors(self):
        """-> (counts[N,2] u16 (dynamic,total), start[N+1] u64, lists u32)."""
        counts = np.zeros((self.n, 2), np.uint16)
        total = self.L.orc_get_neighbor_counts(self.h, _ptr(counts))
        lists = np.zeros(total, np.uint32)
        self.L.orc_get_neighbor_lists(self.h, _ptr(lists))
        start = np.zeros(self.n + 1, np.uint64)
        np.cumsum(counts[:, 1].astype(np.uint64), out=start[1:])
        return counts, start, lists

    def neighbor_flags(self):
        return self.L.orc_get_neighbor_flags(self.h)
