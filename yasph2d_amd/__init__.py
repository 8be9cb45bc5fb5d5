"""yasph2d_amd — MI355X-native DFSPH step loop behind yasph2d's Solver / particle-array surface.

Python is only the test/bench driver here.  The product is libsphx.so (hand-written HIP for gfx950 + a C ABI,
include/sphx.h); the classes below are thin ctypes views of
  * the device solver context (`SphxContext`  — what a Rust `impl Solver` shim would bind), and
  * the C++ host-side mirror of the reference's caller types (`FluidParticleWorld`, `TimeManager`, `DFSPHSolver`),
named after the reference (src/sph/fluidparticleworld.rs, timemanager.rs, solver/dfsph.rs).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (FLAG_DENSITY_ITER_CAP, FLAG_DIVERGENCE_ITER_CAP, FLAG_DENSE_CELL, FLAG_NEIGHBOR_CAP, FLAG_STRAY_PARTICLES, FLAG_WARMUP, KERNEL_POLY6,  # noqa: F401
                   KERNEL_SPIKY, KERNEL_WENDLAND_C2, SphxError, SphxKernelTime, SphxParams, SphxStepStats)

__all__ = ["SphxContext", "FluidParticleWorld", "TimeManager", "DFSPHSolver", "DFSPHMultiSolver", "default_params", "duration_from_secs_f32",
           "duration_as_secs_f32", "SphxError", "WCSPHSolver"]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


LISTS_32BIT = 0xFFFFFFFF  # sphx_params.list_span_limit: never compress the neighbour lists


def default_params(smoothing_factor=2.0, particle_density=10000.0, fluid_density=100.0, device=0, fixed_iterations=(0, 0)):
    """sphx_default_params: the constants of the reference app (main.rs:85-89, dfsph.rs:49-55)."""
    p = SphxParams()
    rc = _lib.lib().sphx_default_params(smoothing_factor, particle_density, fluid_density, C.byref(p))
    if rc:
        raise SphxError(rc, "sphx_default_params")
    p.device = device
    p.fixed_density_iterations, p.fixed_divergence_iterations = fixed_iterations
    return p


def duration_from_secs_f32(secs):
    return _lib.lib().sphx_duration_from_secs_f32(secs)


def duration_as_secs_f32(ns):
    return _lib.lib().sphx_duration_as_secs_f32(ns)


class SphxContext:
    """Device solver context (sphx_ctx).  Mirrors DFSPHSolver + the solver-owned part of FluidParticleWorld."""

    def __init__(self, params=None, **kw):
        self.L = _lib.lib()
        self.params = params if params is not None else default_params(**kw)
        h = C.c_void_p()
        rc = self.L.sphx_create(C.byref(self.params), C.byref(h))
        if rc:
            raise SphxError(rc, self.L.sphx_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self, "_owned", True):
            self.L.sphx_destroy(self.h)
        self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise SphxError(rc, self.L.sphx_last_error(self.h).decode())

    @property
    def n(self):
        return self.L.sphx_num_particles(self.h)

    @property
    def nb(self):
        return self.L.sphx_num_boundary(self.h)

    def set_boundary(self, xy):
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        self._chk(self.L.sphx_set_boundary(self.h, _p(xy), len(xy)))

    def upload(self, pos, vel=None):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        if vel is not None:
            vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 2)
        self._chk(self.L.sphx_upload(self.h, _p(pos), _p(vel), len(pos)))

    def clear_cached(self):
        self._chk(self.L.sphx_clear_cached(self.h))

    def step_begin(self, dt_prev, law=None):
        """Phase A.  law (TimeManager.law(diameter)): the device derives dt itself and starts phase B without waiting for the
        host (sphx_step_begin_law); step_finish then verifies the host's dt against it."""
        v = C.c_float()
        if law is None:
            self._chk(self.L.sphx_step_begin(self.h, dt_prev, C.byref(v)))
        else:
            self._chk(self.L.sphx_step_begin_law(self.h, dt_prev, C.byref(law), C.byref(v)))
        return v.value

    def step_finish(self, dt):
        st = SphxStepStats()
        self._chk(self.L.sphx_step_finish(self.h, dt, C.byref(st)))
        return st.as_dict()

    def view_request(self, stride=1):
        """Start an asynchronous strided download of {x, y, |v|} (the viewer's per-frame data, main.rs:239-258)."""
        n = C.c_uint32()
        self._chk(self.L.sphx_view_request(self.h, stride, C.byref(n)))
        return n.value

    def view_fetch(self, wait=True):
        """-> float32 array [count, 3] (a copy of the pinned buffer), or None if wait=False and the copy is still in flight."""
        ptr, n = C.POINTER(C.c_float)(), C.c_uint32()
        rc = self.L.sphx_view_fetch(self.h, int(wait), C.byref(ptr), C.byref(n))
        if rc == _lib.ERR_NOT_READY and not wait:
            return None
        self._chk(rc)
        if n.value == 0:
            return np.zeros((0, 3), np.float32)
        return np.ctypeslib.as_array(ptr, shape=(n.value, 3)).copy()

    def wcsph_step_begin(self, dt):
        """WCSPHSolver::simulation_step up to the timer call (wscsph.rs:126-161) -> vmax."""
        v = C.c_float()
        self._chk(self.L.sphx_wcsph_step_begin(self.h, dt, C.byref(v)))
        return v.value

    def wcsph_step_finish(self, dt):
        st = SphxStepStats()
        self._chk(self.L.sphx_wcsph_step_finish(self.h, dt, C.byref(st)))
        return st.as_dict()

    def update_neighborhood(self):
        self._chk(self.L.sphx_update_neighborhood(self.h))

    def update_densities(self, kind=KERNEL_WENDLAND_C2):
        self._chk(self.L.sphx_update_densities(self.h, kind))

    def compute_alpha(self):
        self._chk(self.L.sphx_compute_alpha(self.h))

    def synchronize(self):
        self._chk(self.L.sphx_synchronize(self.h))

    def set_tiling_invariant(self, on=True):
        """sphx_set_tiling_invariant: cell mates ordered by persistent id, warm-start values travel with their particle (a comparison
        mode for multi-GPU runs; not the reference's behaviour)."""
        self._chk(self.L.sphx_set_tiling_invariant(self.h, int(bool(on))))

    def download(self, pos=True, vel=True, density=True, ids=True):
        n = self.n
        out = {}
        a_pos = np.zeros((n, 2), np.float32) if pos else None
        a_vel = np.zeros((n, 2), np.float32) if vel else None
        a_den = np.zeros(n, np.float32) if density else None
        a_ids = np.zeros(n, np.uint32) if ids else None
        self._chk(self.L.sphx_download(self.h, _p(a_pos), _p(a_vel), _p(a_den), _p(a_ids)))
        out.update(pos=a_pos, vel=a_vel, density=a_den, ids=a_ids)
        return out

    def download_boundary(self):
        xy = np.zeros((self.nb, 2), np.float32)
        ids = np.zeros(self.nb, np.uint32)
        self._chk(self.L.sphx_download_boundary(self.h, _p(xy), _p(ids)))
        return xy, ids

    def download_solver_state(self):
        n = self.n
        a, k, s = (np.zeros(n, np.float32) for _ in range(3))
        self._chk(self.L.sphx_download_solver_state(self.h, _p(a), _p(k), _p(s)))
        return dict(alpha=a, kappa=k, stiffness=s)

    def download_neighbors(self):
        """-> (counts[N,2] u16 (dynamic,total), start[N+1] u64, lists u32) in the canonical form of sphx.h."""
        n = self.n
        counts = np.zeros((n, 2), np.uint16)
        total = C.c_uint64()
        self._chk(self.L.sphx_download_neighbors(self.h, _p(counts), None, C.byref(total)))
        lists = np.zeros(total.value, np.uint32)
        self._chk(self.L.sphx_download_neighbors(self.h, None, _p(lists), C.byref(total)))
        start = np.zeros(n + 1, np.uint64)
        np.cumsum(counts[:, 1].astype(np.uint64), out=start[1:])
        return counts, start, lists

    def download_cells(self, static=False):
        m = C.c_uint32()
        self._chk(self.L.sphx_download_cells(self.h, int(static), None, None, C.byref(m)))
        first = np.zeros(m.value, np.uint32)
        cidx = np.zeros(m.value, np.uint32)
        self._chk(self.L.sphx_download_cells(self.h, int(static), _p(first), _p(cidx), C.byref(m)))
        return first, cidx

    def last_flags(self):
        return self.L.sphx_last_flags(self.h)

    def grid_info(self, which=0):
        """Cell table behind the grid: covered blocks, table entries, directory extent (sphx_grid_info)."""
        out = (C.c_uint32 * 4)()
        self._chk(self.L.sphx_grid_info(self.h, which, out))
        return dict(blocks=out[0], entries=out[1], nbx=out[2], nby=out[3])

    def constants(self):
        out = np.zeros(6, np.float32)
        self._chk(self.L.sphx_get_constants(self.h, _p(out)))
        return out

    def profile_enable(self, on=True):
        self._chk(self.L.sphx_profile_enable(self.h, int(on)))

    def profile_filter(self, label=None, every=1):
        """Time only every `every`-th launch with this label (None = all launches)."""
        self._chk(self.L.sphx_profile_filter(self.h, label.encode() if label else None, every))

    def profile_reset(self):
        self._chk(self.L.sphx_profile_reset(self.h))

    def profile_event_overhead(self):
        """Mean elapsed ms of an EMPTY hipEvent bracket on the context's stream (what the bracket adds to a timed launch)."""
        v = C.c_double()
        self._chk(self.L.sphx_profile_event_overhead(self.h, C.byref(v)))
        return v.value

    def profile_get(self):
        n = C.c_uint32(0)
        self._chk(self.L.sphx_profile_get(self.h, None, C.byref(n)))
        arr = (SphxKernelTime * max(1, n.value))()
        self._chk(self.L.sphx_profile_get(self.h, arr, C.byref(n)))
        return {arr[i].name.decode(): dict(launches=arr[i].launches, total_ms=arr[i].total_ms, bytes=arr[i].algorithmic_bytes)
                for i in range(n.value)}


class FluidParticleWorld:
    """Host-side world (fluidparticleworld.rs:92-195): scene helpers + the host copies of the particle arrays."""

    def __init__(self, smoothing_factor=2.0, particle_density=10000.0, fluid_density=100.0):
        self.L = _lib.lib()
        self.h = self.L.sphx_world_create(smoothing_factor, particle_density, fluid_density)
        if not self.h:
            raise ValueError("particle_density must be positive")
        self.args = (smoothing_factor, particle_density, fluid_density)

    def close(self):
        if getattr(self, "h", None):
            self.L.sphx_world_destroy(self.h)
            self.h = None

    __del__ = close

    def properties(self):
        out = np.zeros(4, np.float32)
        self.L.sphx_world_properties(self.h, _p(out))
        return dict(smoothing_length=out[0], particle_mass=out[1], particle_radius=out[2], fluid_density=out[3])

    def remove_all_fluid_particles(self):
        self.L.sphx_world_remove_all_fluid_particles(self.h)

    def remove_all_boundary_particles(self):
        self.L.sphx_world_remove_all_boundary_particles(self.h)

    def add_fluid_rect(self, x, y, w, h, jitter_amount):
        self.L.sphx_world_add_fluid_rect(self.h, x, y, w, h, jitter_amount)

    def add_boundary_thick_line(self, start, end, thickness_in_particles):
        self.L.sphx_world_add_boundary_thick_line(self.h, start[0], start[1], end[0], end[1], thickness_in_particles)

    def add_boundary_line(self, start, end):
        self.L.sphx_world_add_boundary_line(self.h, start[0], start[1], end[0], end[1])

    def reset_fluid(self, scale=1.0):
        """main.rs:177-196 dam-break scene, every coordinate multiplied by `scale`."""
        self.L.sphx_world_reset_fluid(self.h, scale)

    def set_particles(self, pos, vel=None):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        if vel is not None:
            vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 2)
        self.L.sphx_world_set_particles(self.h, _p(pos), _p(vel), len(pos))

    def set_boundary(self, xy):
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        self.L.sphx_world_set_boundary(self.h, _p(xy), len(xy))

    @property
    def num_dynamic_particles(self):
        return self.L.sphx_world_num_dynamic_particles(self.h)

    @property
    def num_boundary_particles(self):
        return self.L.sphx_world_num_boundary_particles(self.h)

    def _view(self, ptr, shape, dtype):
        n = int(np.prod(shape))
        if not ptr or n == 0:
            return np.zeros(shape, dtype)
        ct = {np.float32: C.c_float, np.uint32: C.c_uint32}[dtype]
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).reshape(shape).copy()

    @property
    def positions(self):
        return self._view(self.L.sphx_world_positions(self.h), (self.num_dynamic_particles, 2), np.float32)

    @property
    def velocities(self):
        return self._view(self.L.sphx_world_velocities(self.h), (self.num_dynamic_particles, 2), np.float32)

    @property
    def densities(self):
        return self._view(self.L.sphx_world_densities(self.h), (self.num_dynamic_particles,), np.float32)

    @property
    def boundary_particles(self):
        return self._view(self.L.sphx_world_boundary(self.h), (self.num_boundary_particles, 2), np.float32)

    @property
    def particle_ids(self):
        return self._view(self.L.sphx_world_particle_ids(self.h), (self.num_dynamic_particles,), np.uint32)


class TimeManager:
    """timemanager.rs, simulation clock only.  Defaults are the reference app's DFSPH values (main.rs:115-129)."""

    def __init__(self, timestep_max_ns=None, timestep_min_ns=None, cfl_factor=1.5, fixed_ns=None):
        self.L = _lib.lib()
        if fixed_ns is not None:
            self.h = self.L.sphx_timer_create_fixed(fixed_ns)
        else:
            if timestep_max_ns is None:
                timestep_max_ns = duration_from_secs_f32(np.float32(1.0) / np.float32(120.0) / np.float32(3.0))
            if timestep_min_ns is None:
                timestep_min_ns = duration_from_secs_f32(np.float32(1.0) / np.float32(60.0) / np.float32(400.0))
            self.h = self.L.sphx_timer_create_adaptive(timestep_max_ns, timestep_min_ns, cfl_factor)
        self.timestep_max_ns, self.timestep_min_ns, self.cfl_factor = timestep_max_ns, timestep_min_ns, cfl_factor

    def close(self):
        if getattr(self, "h", None):
            self.L.sphx_timer_destroy(self.h)
            self.h = None

    __del__ = close

    def restart(self):
        self.L.sphx_timer_restart(self.h)

    def simulation_step_ns(self):
        return self.L.sphx_timer_simulation_step_ns(self.h)

    def simulation_step(self):
        return duration_as_secs_f32(self.simulation_step_ns())

    def update_simulation_step(self, particle_diameter, max_velocity):
        return self.L.sphx_timer_update_simulation_step(self.h, particle_diameter, max_velocity)

    def set_target_frame(self, target_ns):
        """AdaptiveTimeStepTarget::TargetFrameLength (timemanager.rs:24-36); 0 = None."""
        self.L.sphx_timer_set_target_frame(self.h, target_ns)

    def on_step_started(self):
        """The clock part of simulation_frame_loop (timemanager.rs:244-247): total simulated time advances by the current step."""
        self.L.sphx_timer_on_step_started(self.h)

    def law(self, particle_diameter):
        """sphx_timer_law for SphxContext.step_begin: this timer's config and current step."""
        out = _lib.SphxTimerLaw()
        rc = self.L.sphx_timer_law_of(self.h, particle_diameter, C.byref(out))
        if rc:
            raise SphxError(rc, "sphx_timer_law_of")
        return out

    @property
    def total_simulated_ns(self):
        return self.L.sphx_timer_total_simulated_ns(self.h)

    @property
    def num_steps(self):
        return self.L.sphx_timer_num_steps(self.h)


class DFSPHSolver:
    """Box<dyn Solver> holding the HIP-backed DFSPHSolver (solver/mod.rs:12-18, dfsph.rs:405-526)."""

    _create = "sphx_solver_create_dfsph"

    def __init__(self, world, params=None):
        self.L = _lib.lib()
        h = C.c_void_p()
        rc = getattr(self.L, self._create)(world.h, C.byref(params) if params is not None else None, C.byref(h))
        if rc:
            raise SphxError(rc, self.L.sphx_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.sphx_solver_destroy(self.h)
            self.h = None

    __del__ = close

    def clear_cached_data(self):
        self.L.sphx_solver_clear_cached_data(self.h)

    def simulation_step(self, world, time_manager, sync_world=True):
        st = SphxStepStats()
        rc = self.L.sphx_solver_simulation_step(self.h, world.h, time_manager.h, int(sync_world), C.byref(st))
        if rc:
            raise SphxError(rc, self.L.sphx_solver_last_error(self.h).decode())
        return st.as_dict()

    def simulation_steps(self, world, time_manager, k, sync_world=True):
        """k consecutive simulation_step calls inside the library (the frame loop of main.rs:348-350); returns the k stats dicts."""
        st = (SphxStepStats * k)()
        done = C.c_uint32()
        rc = self.L.sphx_solver_simulation_steps(self.h, world.h, time_manager.h, int(sync_world), k, st, C.byref(done))
        if rc:
            raise SphxError(rc, f"step {done.value} of {k}: " + self.L.sphx_solver_last_error(self.h).decode())
        return [x.as_dict() for x in st]

    def sync_world(self, world):
        rc = self.L.sphx_solver_sync_world(self.h, world.h)
        if rc:
            raise SphxError(rc, self.L.sphx_solver_last_error(self.h).decode())

    def context(self):
        """Borrowed SphxContext view (for inspection: neighbours, cells, solver state, profiling)."""
        ctx = SphxContext.__new__(SphxContext)
        ctx.L = self.L
        ctx.params = None
        ctx._owned = False
        ctx.h = C.c_void_p(self.L.sphx_solver_ctx(self.h))
        return ctx


class DFSPHMultiSolver(DFSPHSolver):
    """The same Box<dyn Solver> over several GPUs (sph::HipDfsphMultiSolver): simulation_step / clear_cached_data / sync_world as before,
    the tiles, the halo exchange and the reductions live inside libsphx (sphx_multi)."""

    def __init__(self, world, devices, params=None, options=None):
        self.L = _lib.lib()
        h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = self.L.sphx_solver_create_dfsph_multi(world.h, C.byref(params) if params is not None else None, devs, len(devices),
                                                   C.byref(options) if options is not None else None, C.byref(h))
        if rc:
            raise SphxError(rc, self.L.sphx_multi_last_error(None).decode())
        self.h = h


class WCSPHSolver(DFSPHSolver):
    """Box<dyn Solver> holding the HIP-backed WCSPHSolver (solver/wscsph.rs); the app pairs it with cfl_factor 0.2 (main.rs:116-119)."""

    _create = "sphx_solver_create_wcsph"
