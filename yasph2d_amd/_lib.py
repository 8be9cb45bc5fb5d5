"""ctypes binding of libsphx.so (include/sphx.h).  Fails loudly if the library is missing: there is no fallback."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPHX_LIB: load another build of the same library (kernel A/B experiments, tools/ab_build.sh)
LIB_PATH = os.environ.get("SPHX_LIB") or os.path.join(_HERE, "libsphx.so")
CSRC = os.path.join(_HERE, "csrc")

# status codes (sphx.h)
OK, ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_READY, ERR_NONFINITE, ERR_NEIGHBOR_PANIC, ERR_CAPACITY, ERR_OUT_OF_DOMAIN = range(9)
FLAG_NEIGHBOR_CAP, FLAG_DENSITY_ITER_CAP, FLAG_DIVERGENCE_ITER_CAP, FLAG_WARMUP, FLAG_STRAY_PARTICLES, FLAG_DENSE_CELL = 1, 2, 4, 8, 16, 32
KERNEL_WENDLAND_C2, KERNEL_POLY6, KERNEL_SPIKY = 0, 1, 2


class SphxParams(C.Structure):
    _fields_ = [
        ("smoothing_length", C.c_float),
        ("particle_mass", C.c_float),
        ("fluid_density", C.c_float),
        ("particle_radius", C.c_float),
        ("gravity", C.c_float * 2),
        ("grid_min", C.c_float * 2),
        ("xsph_epsilon", C.c_float),
        ("max_avg_density_error", C.c_float),
        ("max_density_iterations", C.c_uint32),
        ("max_divergence_error", C.c_float),
        ("max_divergence_iterations", C.c_uint32),
        ("fixed_density_iterations", C.c_uint32),
        ("fixed_divergence_iterations", C.c_uint32),
        ("device", C.c_int32),
        ("list_span_limit", C.c_uint32),
        ("reserved", C.c_uint32 * 3),
    ]


class SphxTimerLaw(C.Structure):
    _fields_ = [
        ("adaptive", C.c_uint32),
        ("cfl_factor", C.c_float),
        ("particle_diameter", C.c_float),
        ("reserved", C.c_uint32),
        ("timestep_min_ns", C.c_uint64),
        ("timestep_max_ns", C.c_uint64),
        ("simulation_step_ns", C.c_uint64),
    ]


class SphxStepStats(C.Structure):
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
        ("flags", C.c_uint32),
        ("remote_entries", C.c_uint32),
        ("neighbor_entries", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class SphxKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double), ("algorithmic_bytes", C.c_double)]


def build(force=False):
    """hipcc --offload-arch=gfx950 build of libsphx.so via csrc/Makefile (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)


class SphxMultiOptions(C.Structure):
    _fields_ = [("halo_cells", C.c_uint32), ("fixed_halo", C.c_uint32), ("rebalance_every", C.c_uint32), ("layout", C.c_uint32),
                ("cap_records", C.c_uint32), ("overlap_exchange", C.c_uint32), ("reserved", C.c_uint32 * 2)]


COMM_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p)
COMM_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_double))
COMM_ABORT = C.CFUNCTYPE(None, C.c_void_p)


class SphxCommOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("exchange", COMM_EXCHANGE), ("allreduce", COMM_ALLREDUCE),
                ("abort", COMM_ABORT)]


class SphxMultiInfo(C.Structure):
    _fields_ = [("world", C.c_uint32), ("local_tiles", C.c_uint32), ("halo_now", C.c_uint32), ("halo_max", C.c_uint32), ("peers", C.c_uint32),
                ("n_local", C.c_uint32), ("cap_records", C.c_uint32), ("grid_layout", C.c_uint32), ("axis", C.c_int32), ("band_packs", C.c_uint32),
                ("exchanges", C.c_uint64), ("rebalances", C.c_uint64), ("build_particles", C.c_uint64), ("neighbor_entries", C.c_uint64),
                ("remote_entries", C.c_uint64), ("owned_local", C.c_uint64), ("transport", C.c_char * 96),
                ("halo_bytes_packed", C.c_uint64), ("halo_bytes_sent", C.c_uint64), ("ownership_seconds", C.c_double)]


LAYOUT_AUTO, LAYOUT_STRIPS, LAYOUT_GRID = 0, 1, 2

# every symbol include/sphx.h declares: (restype, argtypes)
_vp, _u16p = C.c_void_p, C.c_void_p
_f, _u32, _u64, _i = C.c_float, C.c_uint32, C.c_uint64, C.c_int
SIGNATURES = {
    "sphx_abi_version": (_u32, []),
    "sphx_default_params": (_i, [_f, _f, _f, C.POINTER(SphxParams)]),
    "sphx_create": (_i, [C.POINTER(SphxParams), C.POINTER(_vp)]),
    "sphx_destroy": (None, [_vp]),
    "sphx_last_error": (C.c_char_p, [_vp]),
    "sphx_set_boundary": (_i, [_vp, _vp, _u32]),
    "sphx_upload": (_i, [_vp, _vp, _vp, _u32]),
    "sphx_download": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "sphx_download_boundary": (_i, [_vp, _vp, _vp]),
    "sphx_num_particles": (_u32, [_vp]),
    "sphx_num_boundary": (_u32, [_vp]),
    "sphx_clear_cached": (_i, [_vp]),
    "sphx_step_begin": (_i, [_vp, _f, C.POINTER(_f)]),
    "sphx_step_begin_law": (_i, [_vp, _f, _vp, _vp]),
    "sphx_timer_law_of": (_i, [_vp, _f, _vp]),
    "sphx_timer_set_target_frame": (None, [_vp, _u64]),
    "sphx_timer_on_step_started": (None, [_vp]),
    "sphx_step_finish": (_i, [_vp, _f, C.POINTER(SphxStepStats)]),
    "sphx_update_neighborhood": (_i, [_vp]),
    "sphx_update_densities": (_i, [_vp, _i]),
    "sphx_compute_alpha": (_i, [_vp]),
    "sphx_download_solver_state": (_i, [_vp, _vp, _vp, _vp]),
    "sphx_download_neighbors": (_i, [_vp, _vp, _vp, C.POINTER(_u64)]),
    "sphx_download_cells": (_i, [_vp, _i, _vp, _vp, C.POINTER(_u32)]),
    "sphx_grid_info": (_i, [_vp, _i, C.POINTER(_u32)]),
    "sphx_last_flags": (_u32, [_vp]),
    "sphx_get_constants": (_i, [_vp, _vp]),
    "sphx_reserve": (_i, [_vp, _u32]),
    "sphx_tile_configure": (_i, [_vp, _i, _u32, _u32, _u32, _i, _i]),
    "sphx_tile_upload": (_i, [_vp, _vp, _vp, _vp, _u32]),
    "sphx_tile_pack": (_i, [_vp, _vp, _vp, _u32]),
    "sphx_tile_apply": (_i, [_vp, _vp, _vp, _u32]),
    "sphx_tile_count_kept": (_i, [_vp]),
    "sphx_sub_regrid": (_i, [_vp, C.POINTER(_u32)]),
    "sphx_sub_regrid_div": (_i, [_vp, C.POINTER(_u32)]),
    "sphx_sub_regrid_warm": (_i, [_vp, C.POINTER(_u32)]),
    "sphx_sub_nonpressure": (_i, [_vp, _f, C.POINTER(_f)]),
    "sphx_sub_predict": (_i, [_vp, _f]),
    "sphx_sub_warmstart": (_i, [_vp, _i, _f]),
    "sphx_sub_iteration": (_i, [_vp, _i, _f, _i, C.POINTER(C.c_double), C.POINTER(_u64)]),
    "sphx_sub_predict_iteration": (_i, [_vp, _f, C.POINTER(C.c_double), C.POINTER(_u64)]),
    "sphx_tile_band_packs": (_i, [_vp, C.POINTER(_u32)]),
    "sphx_tile_send_counts": (_i, [_vp, _vp, _u32, _u32, C.POINTER(_u32)]),
    "sphx_sub_advect": (_i, [_vp, _f]),
    "sphx_shm_open": (_vp, [C.c_char_p, _i, _i]),
    "sphx_shm_allreduce": (_i, [_vp, _vp, _i, _i, _vp]),
    "sphx_shm_allgather": (_i, [_vp, _vp, _i, _vp]),
    "sphx_shm_abort": (None, [_vp]),
    "sphx_sub_run_ahead": (_i, [_vp, C.c_float]),
    "sphx_tile_carry_warmstart": (_i, [_vp, _i, _i]),
    "sphx_set_tiling_invariant": (_i, [_vp, _i]),
    "sphx_tile_defer_advect": (_i, [_vp, _i]),
    "sphx_build_stats": (_i, [_vp, _vp, _vp, _vp]),
    "sphx_shm_close": (None, [_vp]),
    "sphx_tile_configure_rect": (_i, [_vp, _vp, _u32, _vp, _u32]),
    "sphx_tile_pack_n": (_i, [_vp, _vp, _u32, _u32]),
    "sphx_tile_apply_n": (_i, [_vp, _vp, _u32, _u32]),
    "sphx_tile_advect_pack_n": (_i, [_vp, _f, _vp, _u32, _u32]),
    "sphx_view_request": (_i, [_vp, _u32, C.POINTER(_u32)]),
    "sphx_view_fetch": (_i, [_vp, _i, C.POINTER(C.POINTER(_f)), C.POINTER(_u32)]),
    "sphx_synchronize": (_i, [_vp]),
    "sphx_set_stream": (_i, [_vp, _vp]),
    "sphx_profile_enable": (_i, [_vp, _i]),
    "sphx_profile_reset": (_i, [_vp]),
    "sphx_profile_filter": (_i, [_vp, C.c_char_p, _u32]),
    "sphx_profile_get": (_i, [_vp, _vp, C.POINTER(_u32)]),
    "sphx_profile_event_overhead": (_i, [_vp, C.POINTER(C.c_double)]),
    "sphx_multi_default_options": (_i, [C.POINTER(SphxMultiOptions)]),
    "sphx_multi_create": (_i, [C.POINTER(SphxParams), C.POINTER(C.c_int), _i, C.POINTER(SphxMultiOptions), C.POINTER(_vp)]),
    "sphx_multi_create_rank": (_i, [C.POINTER(SphxParams), _i, C.POINTER(SphxCommOps), C.c_char_p, _i, _i, C.POINTER(SphxMultiOptions), C.POINTER(_vp)]),
    "sphx_multi_destroy": (None, [_vp]),
    "sphx_multi_last_error": (C.c_char_p, [_vp]),
    "sphx_multi_set_layout": (_i, [_vp, _i, _vp, _u32]),
    "sphx_multi_set_grid_layout": (_i, [_vp, _u32, _u32, _vp, _vp]),
    "sphx_multi_set_boundary": (_i, [_vp, _vp, _u32]),
    "sphx_multi_upload": (_i, [_vp, _vp, _vp, _vp, _u32]),
    "sphx_multi_clear_cached": (_i, [_vp]),
    "sphx_multi_step_begin": (_i, [_vp, _f, C.POINTER(_f)]),
    "sphx_multi_step_finish": (_i, [_vp, _f, C.POINTER(SphxStepStats)]),
    "sphx_multi_synchronize": (_i, [_vp]),
    "sphx_multi_num_owned": (_u64, [_vp]),
    "sphx_multi_download": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    "sphx_multi_info": (_i, [_vp, C.POINTER(SphxMultiInfo)]),
    "sphx_multi_tile_ctx": (_vp, [_vp, _u32]),
    "sphx_multi_simulation_step": (_i, [_vp, _vp, _f, C.POINTER(SphxStepStats)]),
    "sphx_multi_simulation_steps": (_i, [_vp, _vp, _f, _u32, C.POINTER(SphxStepStats), C.POINTER(_u32)]),
    # host mirror
    "sphx_world_create": (_vp, [_f, _f, _f]),
    "sphx_world_destroy": (None, [_vp]),
    "sphx_world_properties": (None, [_vp, _vp]),
    "sphx_world_remove_all_fluid_particles": (None, [_vp]),
    "sphx_world_remove_all_boundary_particles": (None, [_vp]),
    "sphx_world_add_fluid_rect": (None, [_vp, _f, _f, _f, _f, _f]),
    "sphx_world_add_boundary_thick_line": (None, [_vp, _f, _f, _f, _f, _u32]),
    "sphx_world_add_boundary_line": (None, [_vp, _f, _f, _f, _f]),
    "sphx_world_reset_fluid": (None, [_vp, _f]),
    "sphx_world_num_dynamic_particles": (_u32, [_vp]),
    "sphx_world_num_boundary_particles": (_u32, [_vp]),
    "sphx_world_positions": (_vp, [_vp]),
    "sphx_world_velocities": (_vp, [_vp]),
    "sphx_world_densities": (_vp, [_vp]),
    "sphx_world_boundary": (_vp, [_vp]),
    "sphx_world_particle_ids": (_vp, [_vp]),
    "sphx_world_set_particles": (None, [_vp, _vp, _vp, _u32]),
    "sphx_world_set_boundary": (None, [_vp, _vp, _u32]),
    "sphx_world_set_gravity": (None, [_vp, _f, _f]),
    "sphx_duration_from_secs_f32": (_u64, [_f]),
    "sphx_duration_as_secs_f32": (_f, [_u64]),
    "sphx_timer_create_adaptive": (_vp, [_u64, _u64, _f]),
    "sphx_timer_create_fixed": (_vp, [_u64]),
    "sphx_timer_destroy": (None, [_vp]),
    "sphx_timer_restart": (None, [_vp]),
    "sphx_timer_simulation_step_ns": (_u64, [_vp]),
    "sphx_timer_update_simulation_step": (_u64, [_vp, _f, _f]),
    "sphx_timer_total_simulated_ns": (_u64, [_vp]),
    "sphx_timer_num_steps": (_u32, [_vp]),
    "sphx_solver_create_dfsph": (_i, [_vp, C.POINTER(SphxParams), C.POINTER(_vp)]),
    "sphx_solver_create_wcsph": (_i, [_vp, C.POINTER(SphxParams), C.POINTER(_vp)]),
    "sphx_solver_create_dfsph_multi": (_i, [_vp, C.POINTER(SphxParams), C.POINTER(C.c_int), _i, C.POINTER(SphxMultiOptions), C.POINTER(_vp)]),
    "sphx_wcsph_step_begin": (_i, [_vp, _f, C.POINTER(_f)]),
    "sphx_wcsph_step_finish": (_i, [_vp, _f, C.POINTER(SphxStepStats)]),
    "sphx_solver_destroy": (None, [_vp]),
    "sphx_solver_clear_cached_data": (None, [_vp]),
    "sphx_solver_simulation_step": (_i, [_vp, _vp, _vp, _i, C.POINTER(SphxStepStats)]),
    "sphx_solver_simulation_steps": (_i, [_vp, _vp, _vp, _i, _u32, C.POINTER(SphxStepStats), C.POINTER(_u32)]),
    "sphx_solver_sync_world": (_i, [_vp, _vp]),
    "sphx_solver_ctx": (_vp, [_vp]),
    "sphx_solver_last_error": (C.c_char_p, [_vp]),
}

_lib = None


def lib():
    """Load libsphx.so.  Raises if it has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (hipcc, gfx950).  yasph2d_amd has no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own HIP runtime.  If libsphx pulled in /opt/rocm's copy first, torch.cuda would later
    # find "No HIP GPUs" in the same process (two runtimes fighting over the device).  Loading torch first makes both share one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class SphxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sphx error {code}: {msg}")
        self.code = code
