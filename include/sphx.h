/* sphx.h — C ABI of the MI355X-native DFSPH step loop (libsphx.so).
 *
 * This is the drop-in boundary behind yasph2d's `Solver` trait / particle-array surface.  Every entry point
 * cites the reference interface (path:line relative to the yasph2d repository root) it replaces.  The Rust-side
 * binding a maintainer would add (an `impl Solver for HipDfsphSolver` in src/sph/solver/) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; `float* xy` arrays are interleaved {x,y} pairs, i.e. exactly the memory of a
 *     `Vec<cgmath::Point2<f32>>` / `Vec<cgmath::Vector2<f32>>` (src/units.rs:2-4).
 *   - every call returns an `int` status (SPHX_OK == 0).  Nothing aborts or throws across the boundary: the places
 *     where the reference panics (dfsph.rs:223,378 `assert!(is_finite)`, neighborhood_search.rs:373 bounds panic,
 *     Duration::from_secs_f32 on non-finite input) become error codes; sphx_last_error() gives the text.
 *   - the caller owns every host pointer (borrowed for the duration of the call); the library owns all device memory.
 *   - one context = one caller thread at a time (mirrors `&mut self` of Solver::simulation_step, solver/mod.rs:17).
 *   - all device work runs on one HIP stream — the context's own, or the caller's (sphx_set_stream); calls that return host data
 *     synchronise it.  The only exception is the viewer feed's device-to-host copy, which has a stream of its own.
 *
 * What is the contract and what is scaffolding (the header has grown beyond the boundary SURVEY.md 8(b) asks for):
 *   STABLE — the drop-in boundary a Rust shim binds:
 *       lifecycle (sphx_default_params, sphx_create, sphx_destroy, sphx_last_error, sphx_abi_version), the particle-array
 *       surface (sphx_set_boundary, sphx_upload, sphx_download*, sphx_num_*, sphx_view_*), the Solver trait (sphx_clear_cached,
 *       sphx_step_begin[_law], sphx_step_finish, sphx_wcsph_step_*), the same trait over a device list (sphx_multi_create[_rank],
 *       sphx_multi_destroy, sphx_multi_set_boundary, sphx_multi_upload, sphx_multi_clear_cached, sphx_multi_step_begin/finish,
 *       sphx_multi_simulation_step[s], sphx_multi_download, sphx_multi_num_owned, sphx_multi_last_error, sphx_comm_ops) and
 *       the status / flag codes.
 *   INSPECTION — parity tests and tools, not on the hot path, may change with the data layout:
 *       sphx_update_neighborhood, sphx_update_densities, sphx_compute_alpha (the pieces benches/ drives), sphx_download_solver_state,
 *       sphx_download_neighbors, sphx_download_cells, sphx_grid_info, sphx_get_constants, sphx_last_flags, sphx_build_stats,
 *       sphx_multi_info, sphx_multi_tile_ctx, sphx_multi_set_layout / _set_grid_layout, sphx_profile_*, sphx_synchronize,
 *       sphx_set_tiling_invariant (a comparison mode: a run that does not depend on how the domain is tiled).
 *   INTERNAL — the seam between the tile loop (csrc/sphx_tiles.cpp) and a tile's context, exported so that the tests can drive the
 *       same sub-steps from the reference implementation of that loop (tests/tiles_reference.py); no stability promise:
 *       sphx_reserve, sphx_tile_*, sphx_sub_*, sphx_set_stream, sphx_shm_*.
 *   HOST MIRROR — the C++ twin of the reference's host types (world, timer, solver object) for hosts without a Rust toolchain:
 *       sphx_world_*, sphx_timer_*, sphx_solver_*, sphx_duration_*.
 */
#ifndef SPHX_H
#define SPHX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPHX_ABI_VERSION 5 /* 5: sphx_shm_allgather, sphx_tile_send_counts, sphx_multi_info_t.halo_bytes_{packed,sent} / .ownership_seconds (appended)
                            * 4: sphx_set_tiling_invariant
                            * 2: sphx_step_stats.remote_entries, sphx_multi_*, frame-loop calls, sphx_sub_regrid_{div,warm}, SPHX_FLAG_DENSE_CELL
                            * 3: sphx_comm_ops.abort, sphx_multi_info_t list statistics, sphx_shm_abort (and sphx_shm_open as a collective),
                            *    sphx_build_stats, sphx_sub_run_ahead, sphx_tile_carry_warmstart, sphx_tile_defer_advect, sphx_sub_predict_iteration, sphx_tile_band_packs,
                            *    sphx_multi_info_t.band_packs (was reserved) */

/* ---- status codes ---- */
enum {
    SPHX_OK = 0,
    SPHX_ERR_INVALID_ARGUMENT = 1,
    SPHX_ERR_NO_DEVICE = 2,       /* no HIP device / HIP runtime error; the product has no CPU fallback */
    SPHX_ERR_HIP = 3,
    SPHX_ERR_NOT_READY = 4,       /* step called before upload / begin-finish out of order */
    SPHX_ERR_NONFINITE = 5,       /* dfsph.rs:223,378 assert!(avg.is_finite()) */
    SPHX_ERR_NEIGHBOR_PANIC = 6,  /* neighborhood_search.rs:373: 64 dynamic neighbours and a static hit (reference panics) */
    SPHX_ERR_CAPACITY = 7,        /* a capacity was exceeded (halo exchange buffer, sphx_reserve, 2^32 table entries) */
    SPHX_ERR_OUT_OF_DOMAIN = 8    /* a particle left the Morton domain the grid tables were sized for */
};

/* ---- stats.flags bits ---- */
enum {
    SPHX_FLAG_NEIGHBOR_CAP = 1u,            /* "particle has too many neighbors" (neighborhood_search.rs:361,376) */
    SPHX_FLAG_DENSITY_ITER_CAP = 2u,        /* "Density error correction canceled" (dfsph.rs:236-245) */
    SPHX_FLAG_DIVERGENCE_ITER_CAP = 4u,     /* "Divergence error correction canceled" (dfsph.rs:391-400) */
    SPHX_FLAG_WARMUP = 8u,                  /* this step ran the warm-up block (dfsph.rs:419-428) */
    SPHX_FLAG_STRAY_PARTICLES = 16u,        /* a particle moved more than a 64-cell block beyond the covered region in one step (a blow-up);
                                               it is kept, without neighbours, and the cell directory is re-covered before the next build */
    SPHX_FLAG_DENSE_CELL = 32u              /* a cell held more than 4096 particles (a collapse to a point, or strays parked together):
                                               their order INSIDE that cell follows arrival, not the previous index — the stable rank
                                               costs occupancy^2 loads and would stall the GPU; everything else is unaffected */
};

/* kernel kinds for sphx_update_densities (src/sph/smoothing_kernel/) */
enum { SPHX_KERNEL_WENDLAND_C2 = 0, SPHX_KERNEL_POLY6 = 1, SPHX_KERNEL_SPIKY = 2 };

typedef struct sphx_ctx sphx_ctx;

/* Everything the reference hard-codes or derives from ConstantFluidProperties, as one POD.
 * sphx_default_params() fills the values of the reference app (main.rs:85-89, dfsph.rs:49-55, xsph.rs:14,
 * fluidparticleworld.rs:123, neighborhood_search.rs:478). */
typedef struct sphx_params {
    float smoothing_length;            /* h: DFSPHSolver::new(_, smoothing_length) dfsph.rs:43; also the search radius / cell size
                                          (fluidparticleworld.rs:118, neighborhood_search.rs:466) */
    float particle_mass;               /* ConstantFluidProperties::particle_mass() fluidparticleworld.rs:74-76 */
    float fluid_density;               /* rho0, fluidparticleworld.rs:70-72 */
    float particle_radius;             /* fluidparticleworld.rs:87-89 (CFL diameter = 2*radius, dfsph.rs:479) */
    float gravity[2];                  /* FluidParticleWorld::gravity, fluidparticleworld.rs:98,123 */
    float grid_min[2];                 /* GridProperties::grid_min, neighborhood_search.rs:478 */
    float xsph_epsilon;                /* XSPHViscosityModel::epsilon, xsph.rs:8,14 */
    float max_avg_density_error;       /* dfsph.rs:49 */
    uint32_t max_density_iterations;   /* dfsph.rs:50 */
    float max_divergence_error;        /* dfsph.rs:53 */
    uint32_t max_divergence_iterations;/* dfsph.rs:54 */
    uint32_t fixed_density_iterations; /* 0 = adaptive (reference behaviour); >0 = run exactly this many (parity/bench mode) */
    uint32_t fixed_divergence_iterations;
    int32_t device;                    /* HIP device ordinal */
    uint32_t list_span_limit;          /* neighbour-list layout (neighborhood_search.rs:262-273, README.md:12 "WIP"): lists are local to a
                                          workgroup of 256 Morton-consecutive particles — 16-bit slots of the record window the traversal
                                          kernels stage in LDS, plus a per-workgroup table of at most this many out-of-window neighbour
                                          entries.  0 = default (512, the table's capacity); SPHX_LISTS_32BIT = 32-bit global indices
                                          everywhere (traversals gather from global memory); smaller values only put more workgroups on
                                          that fallback (test aid).  Results never depend on it. */
    uint32_t reserved[3];
} sphx_params;
#define SPHX_LISTS_32BIT 0xFFFFFFFFu

/* Per-step report (the reference only println!s these; dfsph.rs:227-243,382-398). */
typedef struct sphx_step_stats {
    uint32_t density_iterations;       /* num_density_correction_iterations after the step (dfsph.rs:26) */
    uint32_t divergence_iterations;    /* num_divergence_correction_iterations (dfsph.rs:33) */
    uint32_t warmstart_density;        /* 1 if the kappa warm-start pass ran (dfsph.rs:199-205) */
    uint32_t warmstart_divergence;     /* 1 if the stiffness warm-start pass ran (dfsph.rs:354-360) */
    float avg_density_error;           /* last avg_density_error (dfsph.rs:221) */
    float avg_divergence;              /* last avg_divergence (dfsph.rs:376) */
    float dt_prev;                     /* dt the XSPH term used (dfsph.rs:433) */
    float dt;                          /* dt of prediction/solve/advect (dfsph.rs:478-480) */
    float vmax;                        /* sqrt(max |v + a*dt_prev|^2) handed to TimeManager (dfsph.rs:474-479) */
    uint32_t flags;                    /* SPHX_FLAG_* */
    uint32_t remote_entries;           /* of neighbor_entries: entries outside their workgroup's record window (they go through the
                                          workgroup's out-of-window table; byte accounting of the list layout) */
    uint64_t neighbor_entries;         /* sum of count_total over particles (length of the neighbour list buffer) */
} sphx_step_stats;

/* ---- lifecycle ---------------------------------------------------------------------------------------------- */
/* replaces DFSPHSolver::new (dfsph.rs:43-61) + the solver-owned part of FluidParticleWorld::new (fluidparticleworld.rs:104-127) */
int sphx_default_params(float smoothing_factor, float particle_density, float fluid_density, sphx_params* out);
int sphx_create(const sphx_params* params, sphx_ctx** out_ctx);
void sphx_destroy(sphx_ctx* ctx);
const char* sphx_last_error(const sphx_ctx* ctx); /* ctx may be NULL: returns the last creation error */
uint32_t sphx_abi_version(void);

/* ---- particle-array surface (fluidparticleworld.rs:11-23) ---------------------------------------------------- */
/* Particles::boundary_particles + boundary_changed=true (fluidparticleworld.rs:181-195); rebuilt lazily like :247-252 */
int sphx_set_boundary(sphx_ctx* ctx, const float* xy, uint32_t n);
/* Particles::positions / velocities (vel_xy may be NULL = zero).  Drops nothing cached: like the reference, caches are
 * rebuilt when the particle count differs from the cached arrays' length (dfsph.rs:419) or after sphx_clear_cached. */
int sphx_upload(sphx_ctx* ctx, const float* pos_xy, const float* vel_xy, uint32_t n);
/* Any pointer may be NULL.  Arrays are in the library's current (cell-sorted) order — the reference also re-sorts in
 * place every step (neighborhood_search.rs:121-140).  particle_id[i] = index the particle had in the last sphx_upload. */
int sphx_download(sphx_ctx* ctx, float* pos_xy, float* vel_xy, float* density, uint32_t* particle_id);
int sphx_download_boundary(sphx_ctx* ctx, float* xy, uint32_t* boundary_id);
/* Viewer feed (the app draws every particle at its position, coloured by |v|, main.rs:239-258): {x, y, |v|} of
 * every stride-th particle, packed on the solver stream and copied to pinned host memory on a separate stream, so the transfer
 * overlaps the following steps.  sphx_view_fetch returns the buffer of the latest request (3 floats per entry, valid until the next
 * request); wait = 0 polls (SPHX_ERR_NOT_READY while the copy is in flight). */
int sphx_view_request(sphx_ctx* ctx, uint32_t stride, uint32_t* out_count);
int sphx_view_fetch(sphx_ctx* ctx, int wait, const float** out_xys, uint32_t* out_count);
uint32_t sphx_num_particles(const sphx_ctx* ctx);  /* Particles::num_dynamic_particles  fluidparticleworld.rs:37 */
uint32_t sphx_num_boundary(const sphx_ctx* ctx);   /* Particles::num_boundary_particles fluidparticleworld.rs:41 */

/* ---- Solver trait (solver/mod.rs:12-18) ---------------------------------------------------------------------- */
/* Solver::clear_cached_data (dfsph.rs:406-412) */
int sphx_clear_cached(sphx_ctx* ctx);
/* Solver::simulation_step is two-phase because the reference calls back into the caller-owned TimeManager mid-step:
 *   phase A = dfsph.rs:419-477 (warm-up if needed, non-pressure accelerations + XSPH with dt_prev =
 *             time_manager.simulation_step().as_secs_f32(), max |v + a*dt_prev|) -> *out_vmax = sqrt(max)
 *   host    = dt = time_manager.update_simulation_step(2*radius, vmax).as_secs_f32()   (dfsph.rs:478-480)
 *   phase B = dfsph.rs:484-524 (predict, constant-density loop, advect, re-grid, density+alpha, divergence loop, swap) */
int sphx_step_begin(sphx_ctx* ctx, float dt_prev, float* out_vmax);
int sphx_step_finish(sphx_ctx* ctx, float dt, sphx_step_stats* out_stats);

/* The law the caller's TimeManager is about to apply to vmax (TimeManager::update_simulation_step, timemanager.rs:252-279,
 * with the public TimerConfig of timemanager.rs:10-60 and the current TimeManager::simulation_step()).  Handing it to phase A
 * lets the device derive the same dt right behind its vmax reduction and put the start of phase B (velocity prediction, first
 * constant-density iteration) on the stream before the host has even read vmax: the GPU no longer idles for the host round trip.
 * The host side is unchanged — it still calls update_simulation_step with *out_vmax and passes the result to
 * sphx_step_finish, which verifies bit for bit that both arrived at the same dt (SPHX_ERR_INVALID_ARGUMENT otherwise; the
 * device state is then stale and must be uploaded again). */
typedef struct sphx_timer_law {
    uint32_t adaptive;            /* SimulationStepConfig::Adaptive (1) or ::Fixed (0: dt stays simulation_step_ns) */
    float cfl_factor;             /* AdaptiveTimeStep cfl factor (main.rs:126) */
    float particle_diameter;      /* 2 * particle_radius (dfsph.rs:479) */
    uint32_t reserved;
    uint64_t timestep_min_ns;     /* main.rs:124 */
    uint64_t timestep_max_ns;     /* main.rs:123 */
    uint64_t simulation_step_ns;  /* TimeManager::simulation_step() before the update, in nanoseconds */
} sphx_timer_law;
int sphx_step_begin_law(sphx_ctx* ctx, float dt_prev, const sphx_timer_law* law, float* out_vmax); /* law == NULL: sphx_step_begin */

/* WCSPHSolver::simulation_step (solver/wscsph.rs:126-179), the second Solver behind the same boundary,
 * two-phase for the same reason:
 *   phase A = leap frog 1 with dt = time_manager.simulation_step() (:138-149), update_neighborhood_datastructure (:152),
 *             update_densities(Poly6) (:153), update_accellerations (:59-118, :154), max |v + a*dt| (:158-161) -> *out_vmax
 *   host    = dt = time_manager.update_simulation_step(2*radius, vmax).as_secs_f32()   (:162-164)
 *   phase B = leap frog 2 (:168-177)
 * Constants as WCSPHSolver::new (:31-49): Poly6 density kernel, Spiky pressure kernel, XSPH viscosity, Tait gamma 7,
 * set_compressibility(0.01, 1.0), boundary_force_factor 1.  sphx_clear_cached also drops the accelerations (:122-124).
 * One context runs one solver: the DFSPH and the WCSPH step share the acceleration array. */
int sphx_wcsph_step_begin(sphx_ctx* ctx, float dt, float* out_vmax);
int sphx_wcsph_step_finish(sphx_ctx* ctx, float dt, sphx_step_stats* out_stats);

/* ---- pieces of the path the reference exposes on FluidParticleWorld (driven by benches/) --------------------- */
/* FluidParticleWorld::update_neighborhood_datastructure(vec![], vec![]) (fluidparticleworld.rs:235-261) */
int sphx_update_neighborhood(sphx_ctx* ctx);
/* FluidParticleWorld::update_densities(kernel) (fluidparticleworld.rs:197-231) */
int sphx_update_densities(sphx_ctx* ctx, int kernel_kind);
/* DFSPHSolver::compute_alpha_factors (dfsph.rs:68-97) on the current lists */
int sphx_compute_alpha(sphx_ctx* ctx);

/* ---- parity/inspection (not on the hot path) ------------------------------------------------------------------ */
/* DFSPHSolver::{alpha_values, warmstart_kappa, warmstart_stiffness} (dfsph.rs:36-40); any pointer may be NULL */
int sphx_download_solver_state(sphx_ctx* ctx, float* alpha, float* kappa, float* stiffness);
/* NeighborLists (neighborhood_search.rs:262-300, 433-449) in canonical form: counts[2*i] = count_dynamic,
 * counts[2*i+1] = count_total; lists = all particles' lists concatenated in particle order (the reference's
 * start_index is thread-schedule dependent and is not part of the contract).  Either pointer may be NULL. */
int sphx_download_neighbors(sphx_ctx* ctx, uint16_t* counts, uint32_t* lists, uint64_t* out_total_entries);
/* CompactMortonCellGrid::cells (neighborhood_search.rs:34-37,142-165) incl. the sentinel; which: 0 dynamic, 1 static.
 * Pass NULL arrays to query the count. */
int sphx_download_cells(sphx_ctx* ctx, int which, uint32_t* first_particle, uint32_t* cidx, uint32_t* out_count);
/* SPHX_FLAG_* bits raised since the latest sphx_step_begin (also by the stand-alone sphx_update_neighborhood) */
uint32_t sphx_last_flags(const sphx_ctx* ctx);
/* the cell table behind the grid (which: 0 dynamic, 1 static): out[0] = covered 64x64-cell blocks, out[1] = table
 * entries (= 4096 x blocks: what every build's histogram scan runs over), out[2], out[3] = extent of the block directory */
int sphx_grid_info(const sphx_ctx* ctx, int which, uint32_t* out4);
/* derived kernel constants: out[0..2] = Wendland {h_inv, normalizer, normalizer_grad} (wendland_quintic_c2.rs:24-30),
 * out[3..5] = Poly6 {hsq, normalizer, normalizer_grad} (poly6.rs:16-23) */
int sphx_get_constants(const sphx_ctx* ctx, float* out6);


/* ---- spatial tiles (multi-GPU) ------------------------------------------------------------------------------
 * The reference has no distributed path; these entry points are the device half of the build's own domain decomposition.
 * One context = one tile: the cells with cell_lo <= c < cell_hi along `axis` (0 = x, 1 = y) are OWNED, a halo of `halo_cells`
 * cells on each side holds copies (ghosts) of the neighbours' particles.  The host driver (sphx_multi_* below — csrc/sphx_tiles.cpp; tests/tiles_reference.py is its Python
 * reference implementation) runs the sub-steps below in the order of dfsph.rs:414-525, all-reduces the three
 * per-step scalars, and once per step — between advect and re-grid — exchanges 32-byte halo records with the two spatial
 * neighbours (RCCL send/recv on the device buffers).  With a halo wider than the number of neighbour traversals between two
 * exchanges, ghost values are recomputed locally instead of being exchanged per sub-step.
 * In tile mode the warm-start arrays travel with their particle (a slot-bound array has no meaning across tiles). */
int sphx_reserve(sphx_ctx* ctx, uint32_t capacity); /* device capacity in particles (owned + ghosts + 2 halo buffers); before upload */
int sphx_tile_configure(sphx_ctx* ctx, int axis, uint32_t cell_lo, uint32_t cell_hi, uint32_t halo_cells, int has_left, int has_right);
/* General form ("4 GPUs: 2x2 tiles"): the context owns the cell rectangle *own; peers[] are the rectangles of the
 * tiles that touch it by an edge or a corner (<= SPHX_MAX_TILE_PEERS), in the order of the buffers given to sphx_tile_pack_n /
 * sphx_tile_apply_n.  A strip is a rectangle spanning the whole other axis; sphx_tile_configure/pack/apply are that special case. */
#define SPHX_MAX_TILE_PEERS 8
typedef struct sphx_tile_rect { uint32_t x0, x1, y0, y1; } sphx_tile_rect; /* cells, half-open */
int sphx_tile_configure_rect(sphx_ctx* ctx, const sphx_tile_rect* own, uint32_t halo_cells, const sphx_tile_rect* peers, uint32_t n_peers);
int sphx_tile_pack_n(sphx_ctx* ctx, void* const* d_send, uint32_t n_send, uint32_t cap_records);         /* n_send == n_peers */
int sphx_tile_apply_n(sphx_ctx* ctx, const void* const* d_recv, uint32_t n_recv, uint32_t cap_records);  /* entries may be NULL */
int sphx_tile_advect_pack_n(sphx_ctx* ctx, float dt, void* const* d_send, uint32_t n_send, uint32_t cap_records); /* sphx_sub_advect + sphx_tile_pack_n in one pass */
int sphx_tile_upload(sphx_ctx* ctx, const float* pos_xy, const float* vel_xy, const uint32_t* ids, uint32_t n); /* owned particles, global ids < 2^31 */
#define SPHX_HALO_RECORD_BYTES 32 /* {float4 pos+vel, u32 id, f32 kappa, f32 stiffness, u32 pad}; record 0 = header (count in .id) */
int sphx_tile_pack(sphx_ctx* ctx, void* d_send_left, void* d_send_right, uint32_t cap_records);              /* DEVICE buffers, (1+cap)*32 B */
int sphx_tile_apply(sphx_ctx* ctx, const void* d_from_left, const void* d_from_right, uint32_t cap_records); /* DEVICE buffers or NULL */
int sphx_tile_count_kept(sphx_ctx* ctx); /* between pack and apply: cell count of the kept particles, overlapping the exchange */
int sphx_sub_regrid(sphx_ctx* ctx, uint32_t* out_n_local);                 /* dfsph.rs:512-518 on owned + ghosts */
/* The same, and the neighbour build also does the first compute_density_change (dfsph.rs:249-280) of the divergence loop that
 * follows: the next sphx_sub_iteration(divergence = 1, first = 1) then skips that pass.  Only for a loop that starts WITHOUT a
 * warm start (sphx_sub_warmstart after this call is refused: the pass has already zeroed the warm-start values). */
int sphx_sub_regrid_div(sphx_ctx* ctx, uint32_t* out_n_local);
/* The other case: the divergence loop that follows starts WITH a warm start (dfsph.rs:354-360); the neighbour build applies it, and
 * the sphx_sub_warmstart(divergence = 1) call that follows returns without launching anything. */
int sphx_sub_regrid_warm(sphx_ctx* ctx, uint32_t* out_n_local);
/* The tile loop's run-ahead over the step boundary: arms the NEXT sphx_sub_iteration to queue the next step's non-pressure pass (with
 * dt_prev = dt_prev_of_next_step) and the publish of its maximum behind its own kernels; the next sphx_sub_nonpressure with the same
 * dt_prev adopts the result if nothing has touched the context in between, and simply runs again otherwise. */
int sphx_sub_run_ahead(sphx_ctx* ctx, float dt_prev_of_next_step);
/* Tile mode: which warm-start arrays (warmstart_kappa, warmstart_stiffness) the re-grids move with their particles (default: both).
 * An array the next solver loop zeroes before it reads it (no warm start: dfsph.rs:199 / :354) need not travel. */
int sphx_tile_carry_warmstart(sphx_ctx* ctx, int kappa, int stiffness);
/* Tile mode, for callers that run sphx_tile_advect_pack_n -> exchange -> sphx_tile_apply_n -> sphx_sub_regrid* back to back: the packing
 * pass classifies, counts and sends the advected particles but leaves the records of the particles the tile keeps to the re-grid's
 * gather, which applies the same x += v* dt while it moves them (24 bytes per particle less in the packing pass).  Any other entry
 * point that looks at the records in between applies the pending advection first. */
int sphx_tile_defer_advect(sphx_ctx* ctx, int on);
/* Tile mode statistics: halo packs (sphx_tile_advect_pack_n) that found their particles classified by the density loop's last
 * correction — send counts per workgroup, kept / retired, cell count (SPHX_TILE_FUSE_CLASS=0 turns that off) — and only visited
 * the workgroups that send something. */
int sphx_tile_band_packs(const sphx_ctx* ctx, uint32_t* out);
/* The record counts in the headers of the send buffers the last sphx_tile_pack_n / sphx_tile_advect_pack_n filled (clamped to
 * cap_records), on the host: waits for the packing kernels (one small device-to-host copy per peer).  The caller may then move
 * (1 + out_counts[k]) * 32 bytes per peer instead of the buffers' capacity. */
int sphx_tile_send_counts(sphx_ctx* ctx, void* const* d_send, uint32_t n_send, uint32_t cap_records, uint32_t* out_counts);
int sphx_sub_nonpressure(sphx_ctx* ctx, float dt_prev, float* out_vmax_sq); /* dfsph.rs:436-477; max over OWNED particles */
int sphx_sub_predict(sphx_ctx* ctx, float dt);                             /* dfsph.rs:484-492 */
int sphx_sub_warmstart(sphx_ctx* ctx, int divergence, float dt);           /* dfsph.rs:199-205 / :354-360 */
int sphx_sub_iteration(sphx_ctx* ctx, int divergence, float dt, int first, double* out_err_sum, uint64_t* out_n_owned); /* :217-221 / :372-377 */
int sphx_sub_advect(sphx_ctx* ctx, float dt);                              /* dfsph.rs:499-510 */
/* sphx_sub_predict followed by sphx_sub_iteration(divergence = 0, first = 1) in one call, for a density loop that starts without a
 * warm start (dfsph.rs:199): one list walk does both (the prediction costs no pass of its own).  Same results as the two calls. */
int sphx_sub_predict_iteration(sphx_ctx* ctx, float dt, double* out_err_sum, uint64_t* out_n_owned);


/* ---- multi-GPU solver behind the Solver boundary ("device list ... internally may drive 1-8 GPUs") ----------------
 * The reference's caller holds ONE Box<dyn Solver> (main.rs:50) and calls simulation_step(&mut world, &mut time_manager)
 * (solver/mod.rs:12-18, main.rs:279).  sphx_multi is that one object over several GPUs: it cuts the domain into tiles (strips along
 * the longer side; 2 x N/2 rectangles on 4 tiles), owns one context per tile and runs the whole tile step loop — ring-budget halo,
 * one exchange per step, adaptive band, re-partitioning — inside the library.  Same two-phase step as sphx_step_begin/finish. */
typedef struct sphx_multi sphx_multi;
enum { SPHX_LAYOUT_AUTO = 0, SPHX_LAYOUT_STRIPS = 1, SPHX_LAYOUT_GRID = 2 };
typedef struct sphx_multi_options {
    uint32_t halo_cells;       /* widest ghost band in cells (16); the band in use follows the ring budget unless fixed_halo */
    uint32_t fixed_halo;       /* 1: always exchange the full band */
    uint32_t rebalance_every;  /* steps between re-partitions of the cuts (16; 0 = never) */
    uint32_t layout;           /* SPHX_LAYOUT_* (auto: 2x2 on 4 tiles, strips otherwise) */
    uint32_t cap_records;      /* records per halo buffer (0 = estimated from the uploaded scene) */
    uint32_t overlap_exchange; /* 1: halo records on a second stream, the re-grid's cell count of the kept particles meanwhile */
    uint32_t reserved[2];
} sphx_multi_options;
/* Communicator supplied by the caller for one tile of a multi-process run (NULL: the built-in one — grouped ncclSend/ncclRecv over
 * RCCL for the halo records, the shared-memory all-reduce below for the scalars).  exchange: send d_send[k] to rank peers[k] and
 * receive d_recv[k] from it, `bytes` each, DEVICE buffers, ordered on hip_stream (no host synchronisation required of the caller's
 * caller).  allreduce: op 0 = sum, 1 = max over n <= 8 doubles; every rank must receive the same bits. */
typedef struct sphx_comm_ops {
    void* user;
    int rank, world;
    int (*exchange)(void* user, const int* peers, int n_peers, void* const* d_send, void* const* d_recv, size_t bytes, void* hip_stream);
    int (*allreduce)(void* user, const double* in, int n, int op, double* out);
    /* optional (may be NULL): this rank has failed and will not take part in further calls — release the ranks waiting for it */
    void (*abort)(void* user);
} sphx_comm_ops;
typedef struct sphx_multi_info_t {
    uint32_t world, local_tiles, halo_now, halo_max, peers, n_local, cap_records, grid_layout;
    int32_t axis;
    uint32_t band_packs; /* of tile 0's halo exchanges: how many packed from the classification its last density correction had made
                            (only the workgroups inside a send band were visited again) */
    uint64_t exchanges, rebalances;
    /* the local tiles' latest neighbour build (measured, not assumed): particles it ran over (owned + ghosts), list entries, and
     * entries outside the workgroup windows — mean list length = neighbor_entries / build_particles */
    uint64_t build_particles, neighbor_entries, remote_entries;
    uint64_t owned_local; /* particles the local tiles own */
    char transport[96];
    /* tile 0's halo exchanges so far, bytes it sent to all its peers together: what its packing pass filled ((1 + records) * 32 per
     * peer) and what travelled (the same rounded up to 64 KiB when the record counts were exchanged first, else the buffers' capacity) */
    uint64_t halo_bytes_packed, halo_bytes_sent;
    double ownership_seconds; /* set-up, tile 0: cell, owner and send-band count of every particle of the global scene (host threads) */
} sphx_multi_info_t;
int sphx_multi_default_options(sphx_multi_options* out);
/* all tiles in this process: tile r runs on HIP device devices[r] (a device may appear more than once); one host thread per tile */
int sphx_multi_create(const sphx_params* params, const int* devices, int n_devices, const sphx_multi_options* opt, sphx_multi** out);
/* ONE tile (rank of world) of a one-process-per-GPU run; comm == NULL: built-in RCCL + shared memory, `job` names the shared segment
 * (unique per run, e.g. the rendezvous port) */
int sphx_multi_create_rank(const sphx_params* params, int device, const sphx_comm_ops* comm, const char* job, int rank, int world,
                           const sphx_multi_options* opt, sphx_multi** out);
void sphx_multi_destroy(sphx_multi* m);
const char* sphx_multi_last_error(const sphx_multi* m); /* m may be NULL: the last creation error */
/* optional explicit cuts (cells) instead of the particle-count quantiles of the uploaded scene: strips along axis, or nx columns
 * (xcuts[nx+1]) each cut again at its own ycuts[ix*(ny+1) ..] */
int sphx_multi_set_layout(sphx_multi* m, int axis, const uint32_t* cuts, uint32_t n_cuts);
int sphx_multi_set_grid_layout(sphx_multi* m, uint32_t nx, uint32_t ny, const uint32_t* xcuts, const uint32_t* ycuts);
int sphx_multi_set_boundary(sphx_multi* m, const float* xy, uint32_t n);  /* the GLOBAL boundary; every tile clips its part */
/* the GLOBAL particle arrays (every rank of a multi-process run passes the same ones; ids == NULL: 0..n-1); includes the warm-up
 * block of dfsph.rs:419-428 */
int sphx_multi_upload(sphx_multi* m, const float* pos_xy, const float* vel_xy, const uint32_t* ids, uint32_t n);
int sphx_multi_clear_cached(sphx_multi* m);                                           /* Solver::clear_cached_data */
int sphx_multi_step_begin(sphx_multi* m, float dt_prev, float* out_vmax);             /* dfsph.rs:419-477, vmax over ALL tiles */
int sphx_multi_step_finish(sphx_multi* m, float dt, sphx_step_stats* out_stats);      /* dfsph.rs:484-524 */
int sphx_multi_synchronize(sphx_multi* m);
uint64_t sphx_multi_num_owned(const sphx_multi* m);
/* owned particles of the local tiles (in-process: all particles), tile after tile; *inout_n: capacity in, count out */
int sphx_multi_download(sphx_multi* m, float* pos_xy, float* vel_xy, float* density, uint32_t* ids, uint64_t* inout_n);
int sphx_multi_info(const sphx_multi* m, sphx_multi_info_t* out);
sphx_ctx* sphx_multi_tile_ctx(sphx_multi* m, uint32_t local_tile); /* inspection (neighbours, cells, profiling) */
/* Solver::simulation_step(&mut world, &mut time_manager) (solver/mod.rs:17) in ONE call: phase A, the TimeManager mirror
 * (simulation_step() dfsph.rs:433, update_simulation_step dfsph.rs:478-480), phase B */
struct sphx_timer;
int sphx_multi_simulation_step(sphx_multi* m, struct sphx_timer* timer, float particle_diameter, sphx_step_stats* out_stats);
/* `k` of those back to back — the caller's frame loop (main.rs:348-350 -> single_sim_step, :279) on this side of the boundary, for
 * hosts whose per-call cost matters (a Python driver).  out_stats: k entries or NULL; *out_done (may be NULL) = steps finished;
 * stops at the first failing step and returns its code. */
int sphx_multi_simulation_steps(sphx_multi* m, struct sphx_timer* timer, float particle_diameter, uint32_t k, sphx_step_stats* out_stats,
                                uint32_t* out_done);

/* ---- single-node scalar reductions through POSIX shared memory ---------------------------------------------------------------
 * The three per-step scalars of the tile driver (vmax, two residual sums) already sit in host memory (pinned mailbox) on every
 * rank; for one process per GPU on ONE node a shared-memory all-reduce costs ~1 us instead of a device round trip through
 * RCCL.  RCCL is used where the path really exchanges data (the halo records).  name: unique per job (e.g. MASTER_PORT). */
typedef struct sphx_shm sphx_shm;
/* Collective: every rank of the run calls it (in any order); it returns once all `world` ranks have joined the SAME segment — rank 0
 * replaces whatever an earlier run left under the name, a rank that attached to such a leftover notices (nobody answers its join
 * token) and attaches again.  NULL on failure or when a rank does not show up within min(SPHX_SHM_TIMEOUT_S, 120) seconds. */
sphx_shm* sphx_shm_open(const char* name, int rank, int world);
/* op: 0 = sum, 1 = max; n <= 8 doubles; every rank gets the same bits (ranks are combined in rank order).  Returns
 * SPHX_ERR_NOT_READY — on every waiting rank, at once — when a rank has called sphx_shm_abort / sphx_shm_close instead of arriving, or
 * after SPHX_SHM_TIMEOUT_S seconds (default 300) without it. */
int sphx_shm_allreduce(sphx_shm* h, const double* in, int n, int op, double* out);
/* every rank's n <= 8 doubles to every rank: out[r * n + k] = rank r's in[k] (world * n doubles).  Same failure behaviour.  The tile
 * driver publishes the record counts of its halo messages this way, so that ncclSend / ncclRecv move what was packed, not the
 * buffers' capacity. */
int sphx_shm_allgather(sphx_shm* h, const double* in, int n, double* out);
void sphx_shm_abort(sphx_shm* h); /* this rank has failed: release the ranks that wait for it */
void sphx_shm_close(sphx_shm* h);

/* NOT the reference's behaviour — a comparison mode for multi-GPU runs.  Two things in a DFSPH run depend on how the domain is cut into
 * tiles: the order of the particles inside a cell (the reference's par_sort_unstable_by_key leaves it open, neighborhood_search.rs:118;
 * this build keeps them in the order of their previous index, which a tile that appends what it receives cannot reproduce) and the
 * warm-start values, which the reference leaves bound to their slot when the particles are re-sorted (dfsph.rs:512) — a slot means
 * nothing across tiles, so tiles let them travel with the particle.  With this switch a context orders the particles of a cell by their
 * persistent id (sphx_download's particle_id) and moves the warm-start values with them: a single context, any sphx_multi tiling and
 * the oracle in the same mode then compute the same run (tests/test_gpu_tiles_full.py, tests/test_gpu_multi.py at 64 M / 128 M).  Off by
 * default; tile contexts take the switch for the cell order (their warm-start values always travel).  Two exceptions: particles with
 * EQUAL ids (caller-supplied through sphx_multi_upload) keep their previous order among themselves, and a cell with more than 4 096
 * particles (a collapse to a point; SPHX_FLAG_DENSE_CELL) keeps arrival order — for those the run may depend on the tiling. */
int sphx_set_tiling_invariant(sphx_ctx* ctx, int on);

/* ---- measurement ---------------------------------------------------------------------------------------------- */
int sphx_synchronize(sphx_ctx* ctx);
/* the latest neighbour build of this context: particles it ran over, list entries in total, entries outside the workgroup windows */
int sphx_build_stats(const sphx_ctx* ctx, uint32_t* out_particles, uint64_t* out_entries, uint64_t* out_remote);
/* Run this context on a HIP stream owned by the caller (hipStream_t; NULL = back to a private stream).  The tile driver passes
 * the stream RCCL orders its sends/receives against, so packing, exchange and unpacking need no host synchronisation. */
int sphx_set_stream(sphx_ctx* ctx, void* hip_stream);
/* When enabled every kernel launch is bracketed by hipEvents on the context's stream; totals are kept per kernel name. */
int sphx_profile_enable(sphx_ctx* ctx, int on);
int sphx_profile_reset(sphx_ctx* ctx);
int sphx_profile_filter(sphx_ctx* ctx, const char* label, uint32_t every); /* time only every `every`-th launch with this label (NULL = all
                                                                              launches): light enough for a timed region */
/* Fills up to *inout_n records; names are NUL-terminated, <= 47 chars. */
typedef struct sphx_kernel_time {
    char name[48];
    uint64_t launches;
    double total_ms;
    double algorithmic_bytes; /* sum over launches of the algorithmic byte count of DESIGN.md */
} sphx_kernel_time;
int sphx_profile_get(sphx_ctx* ctx, sphx_kernel_time* out, uint32_t* inout_n);
/* what the hipEvent bracket itself adds to a measured launch: mean elapsed time between the two events of an EMPTY bracket */
int sphx_profile_event_overhead(sphx_ctx* ctx, double* out_ms);

/* ======================================================================================================================
 * Host-side mirror of the reference's caller-side types (scene helpers, TimeManager, Solver object).  These exist so the
 * C++ harness / Python drivers can play the role of the Rust application; a Rust host would keep using its own types.
 * ====================================================================================================================== */
typedef struct sphx_world sphx_world;   /* FluidParticleWorld (fluidparticleworld.rs:92-102) host arrays + properties */
typedef struct sphx_timer sphx_timer;   /* TimeManager (timemanager.rs:72-92), simulation-step part only */
typedef struct sphx_solver sphx_solver; /* Box<dyn Solver> (main.rs:50): HIP-backed DFSPHSolver */

sphx_world* sphx_world_create(float smoothing_factor, float particle_density, float fluid_density); /* fluidparticleworld.rs:104 */
void sphx_world_destroy(sphx_world* w);
void sphx_world_properties(const sphx_world* w, float* out4); /* {smoothing_length, particle_mass, particle_radius, fluid_density} */
void sphx_world_remove_all_fluid_particles(sphx_world* w);    /* fluidparticleworld.rs:129-132 */
void sphx_world_remove_all_boundary_particles(sphx_world* w); /* fluidparticleworld.rs:134-137 */
void sphx_world_add_fluid_rect(sphx_world* w, float x, float y, float width, float height, float jitter_amount); /* :140-166 */
void sphx_world_add_boundary_thick_line(sphx_world* w, float sx, float sy, float ex, float ey, uint32_t thickness); /* :168-179 */
void sphx_world_add_boundary_line(sphx_world* w, float sx, float sy, float ex, float ey); /* :181-195 */
/* main.rs:177-196 `reset_fluid` with every coordinate multiplied by `scale` (scale 1 = the reference scene, ~4050 particles) */
void sphx_world_reset_fluid(sphx_world* w, float scale);
uint32_t sphx_world_num_dynamic_particles(const sphx_world* w);  /* fluidparticleworld.rs:37 */
uint32_t sphx_world_num_boundary_particles(const sphx_world* w); /* fluidparticleworld.rs:41 */
float* sphx_world_positions(sphx_world* w);   /* interleaved xy, length 2*num_dynamic */
float* sphx_world_velocities(sphx_world* w);
float* sphx_world_densities(sphx_world* w);
float* sphx_world_boundary(sphx_world* w);
uint32_t* sphx_world_particle_ids(sphx_world* w); /* valid after a solver step with sync enabled */
void sphx_world_set_particles(sphx_world* w, const float* pos_xy, const float* vel_xy, uint32_t n);
void sphx_world_set_boundary(sphx_world* w, const float* xy, uint32_t n);
void sphx_world_set_gravity(sphx_world* w, float gx, float gy);

uint64_t sphx_duration_from_secs_f32(float secs); /* std::time::Duration::from_secs_f32 -> nanoseconds (round-to-nearest-even) */
float sphx_duration_as_secs_f32(uint64_t nanos);  /* Duration::as_secs_f32 */
sphx_timer* sphx_timer_create_adaptive(uint64_t timestep_max_ns, uint64_t timestep_min_ns, float cfl_factor); /* timemanager.rs:44-58,105-129 */
sphx_timer* sphx_timer_create_fixed(uint64_t timestep_ns);                                                     /* timemanager.rs:40 */
void sphx_timer_destroy(sphx_timer* t);
void sphx_timer_restart(sphx_timer* t);                                                     /* timemanager.rs:131-133 */
uint64_t sphx_timer_simulation_step_ns(const sphx_timer* t);                                /* timemanager.rs:136-138 */
uint64_t sphx_timer_update_simulation_step(sphx_timer* t, float particle_diameter, float max_velocity); /* timemanager.rs:252-279 */
uint64_t sphx_timer_total_simulated_ns(const sphx_timer* t);
uint32_t sphx_timer_num_steps(const sphx_timer* t);
int sphx_timer_law_of(const sphx_timer* t, float particle_diameter, sphx_timer_law* out); /* fills sphx_timer_law from the mirror */
void sphx_timer_set_target_frame(sphx_timer* t, uint64_t target_ns); /* AdaptiveTimeStepTarget::TargetFrameLength, timemanager.rs:24-36; 0 = None */
void sphx_timer_on_step_started(sphx_timer* t);                      /* the clock part of simulation_frame_loop, timemanager.rs:244-247 */

/* DFSPHSolver::new(XSPHViscosityModel::new(h), h) boxed as dyn Solver (main.rs:93-101).  `params` may be NULL (defaults from the world). */
int sphx_solver_create_dfsph(const sphx_world* w, const sphx_params* params, sphx_solver** out);
int sphx_solver_create_wcsph(const sphx_world* w, const sphx_params* params, sphx_solver** out); /* WCSPHSolver::new, wscsph.rs:29-42 */
/* the same Box<dyn Solver> over several GPUs (one tile per entry of devices[]; sphx_multi inside): the caller's loop does not change */
int sphx_solver_create_dfsph_multi(const sphx_world* w, const sphx_params* params, const int* devices, int n_devices,
                                   const sphx_multi_options* options, sphx_solver** out);
void sphx_solver_destroy(sphx_solver* s);
void sphx_solver_clear_cached_data(sphx_solver* s); /* Solver::clear_cached_data */
/* Solver::simulation_step(&mut world, &mut time_manager) (dfsph.rs:414).  sync_world != 0 copies positions/velocities/
 * densities back into the host world before returning (what main.rs:242-258 draws from); 0 keeps them device-resident. */
int sphx_solver_simulation_step(sphx_solver* s, sphx_world* w, sphx_timer* t, int sync_world, sphx_step_stats* out_stats);
/* `k` consecutive simulation_step calls — the frame loop of main.rs:348-350 (PerformStepAndCallAgain -> single_sim_step, :279) on
 * this side of the boundary, for hosts whose per-call cost matters (a Python driver: ~8 us a step at 1 M particles).  Nothing is
 * skipped or batched on the device: it IS the loop `for _ in 0..k { solver.simulation_step(world, timer) }`.  out_stats: k entries
 * or NULL; *out_done (may be NULL) = steps finished; stops at the first failing step and returns its code. */
int sphx_solver_simulation_steps(sphx_solver* s, sphx_world* w, sphx_timer* t, int sync_world, uint32_t k, sphx_step_stats* out_stats,
                                 uint32_t* out_done);
int sphx_solver_sync_world(sphx_solver* s, sphx_world* w); /* explicit download into the host world */
sphx_ctx* sphx_solver_ctx(sphx_solver* s);
const char* sphx_solver_last_error(const sphx_solver* s);

#ifdef __cplusplus
}
#endif
#endif /* SPHX_H */
