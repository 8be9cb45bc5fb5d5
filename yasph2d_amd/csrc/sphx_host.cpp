// sphx_host.cpp — implementation of the host-side mirror (sphx_host.hpp) and its C exports (include/sphx.h, bottom half).
#include "sphx_host.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace sph {

static inline Vector operator-(Point a, Point b) { return Vector{a.x - b.x, a.y - b.y}; }
static inline Point operator+(Point a, Vector b) { return Point{a.x + b.x, a.y + b.y}; }
static inline Vector operator+(Vector a, Vector b) { return Vector{a.x + b.x, a.y + b.y}; }
static inline Vector operator*(Vector a, Real s) { return Vector{a.x * s, a.y * s}; }
static inline Vector operator/(Vector a, Real s) { return Vector{a.x / s, a.y / s}; }
static inline Vector operator-(Vector a) { return Vector{-a.x, -a.y}; }

// ---- Duration ---------------------------------------------------------------------------------------------------
Duration Duration::from_secs_f32(Real secs) {
    Duration d;
    if (!(secs >= 0.0f) || std::isinf(secs)) return d;  // the reference panics; callers check finiteness first
    uint32_t bits;
    std::memcpy(&bits, &secs, 4);
    const uint32_t bexp = (bits >> 23) & 0xFF;
    uint64_t mant = bits & 0x7FFFFFu;
    int exp2;  // secs == mant * 2^exp2 exactly
    if (bexp == 0) {
        exp2 = -149;
    } else {
        mant |= 0x800000u;
        exp2 = (int)bexp - 150;
    }
    const unsigned __int128 num = (unsigned __int128)mant * 1000000000ull;
    if (exp2 >= 0) {
        d.ns = (uint64_t)(num << exp2);
        return d;
    }
    const int sh = -exp2;
    if (sh >= 100) return d;
    const unsigned __int128 q = num >> sh;
    const unsigned __int128 rem = num - (q << sh);
    const unsigned __int128 half = (unsigned __int128)1 << (sh - 1);
    uint64_t ns = (uint64_t)q;
    if (rem > half || (rem == half && (ns & 1))) ns += 1;
    d.ns = ns;
    return d;
}
Real Duration::as_secs_f32() const {
    const uint64_t secs = ns / 1000000000ull;
    const uint32_t nanos = (uint32_t)(ns % 1000000000ull);
    return (Real)secs + (Real)nanos / 1000000000.0f;
}

// ---- ConstantFluidProperties (fluidparticleworld.rs:52-90) ---------------------------------------------------------
ConstantFluidProperties::ConstantFluidProperties(Real smoothing_factor, Real particle_density, Real fluid_density) {
    smoothing_length_ = 2.0f * particle_radius_from_particle_density(particle_density) * smoothing_factor;
    particle_density_ = particle_density;
    fluid_density_ = fluid_density;
}
Real ConstantFluidProperties::num_particles_per_meter() const { return std::sqrt(particle_density_); }
Real ConstantFluidProperties::particle_radius_from_particle_density(Real particle_density) { return 0.5f / std::sqrt(particle_density); }

// ---- jitter generator ---------------------------------------------------------------------------------------------
// The reference seeds rand 0.8 SmallRng with the particle count before the add (fluidparticleworld.rs:153) and draws a
// cgmath Vector2 of two Standard f32 in [0,1).  SmallRng's stream is not reproducible without the crate (not vendored,
// no lockfile), so the mirror defines its own generator: SplitMix64 seeded the same way, f32 = (top 24 bits) * 2^-24
// (the bit recipe rand's Standard uses for f32).  Initial positions are INPUTS to the path, not part of its parity.
struct SmallRngStandIn {
    uint64_t state;
    explicit SmallRngStandIn(uint64_t seed) : state(seed) {}
    uint64_t next_u64() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    Real gen_f32() { return (Real)(uint32_t)(next_u64() >> 40) * (1.0f / 16777216.0f); }
    Vector gen_vector() {
        const Real x = gen_f32();
        const Real y = gen_f32();
        return Vector{x, y};
    }
};

// ---- FluidParticleWorld ---------------------------------------------------------------------------------------------
FluidParticleWorld::FluidParticleWorld(Real smoothing_factor, Real particle_density, Real fluid_density)
    : properties(smoothing_factor, particle_density, fluid_density), gravity{0.0f, -9.81f}, boundary_changed(true) {}

void FluidParticleWorld::remove_all_fluid_particles() {  // :129-132
    particles.positions.clear();
    particles.velocities.clear();
    stale_prefix = 0;
    fluid_generation++;
}
void FluidParticleWorld::remove_all_boundary_particles() {  // :134-137 (the reference also clears velocities here)
    particles.boundary_particles.clear();
    particles.velocities.clear();
    boundary_changed = true;
    fluid_generation++;
}

void FluidParticleWorld::add_fluid_rect(Real rx, Real ry, Real rw, Real rh, Real jitter_amount) {  // :140-166
    const Real num_particles_per_meter = properties.num_particles_per_meter() * 0.9f;
    const size_t num_particles_x = std::max<size_t>(1, (size_t)(rw * num_particles_per_meter));
    const size_t num_particles_y = std::max<size_t>(1, (size_t)(rh * num_particles_per_meter));
    const size_t num_particles = num_particles_x * num_particles_y;
    const size_t new_total = particles.positions.size() + num_particles;
    particles.positions.reserve(new_total);
    particles.velocities.resize(new_total, Vector{0, 0});
    particles.densities.resize(new_total, 0.0f);

    SmallRngStandIn rng((uint64_t)particles.positions.size());
    const Point bottom_left{rx, ry};
    const Real step = std::min(rw / (Real)num_particles_x, rh / (Real)num_particles_y);
    const Real jitter_factor = step * jitter_amount;
    for (size_t y = 0; y < num_particles_y; ++y) {
        for (size_t x = 0; x < num_particles_x; ++x) {
            const Vector jitter = (rng.gen_vector() * 0.5f + Vector{0.5f, 0.5f}) * jitter_factor;
            particles.positions.push_back(bottom_left + jitter + Vector{step * (Real)x, step * (Real)y});
        }
    }
    fluid_generation++;
}

void FluidParticleWorld::add_boundary_thick_line(Point start, Point end, uint32_t thickness_in_particles) {  // :168-179
    const Vector d = end - start;
    const Real mag = std::sqrt(d.x * d.x + d.y * d.y);
    const Vector dir = d * (1.0f / mag);  // cgmath normalize() = self * (1 / magnitude)
    const Vector dir_perpendicular{-dir.y, dir.x};
    const Real thickness_world = (Real)thickness_in_particles / properties.num_particles_per_meter();
    const Vector elongation = dir * thickness_world;
    Vector offset = (-dir_perpendicular) * thickness_world;
    const Vector step = dir_perpendicular * thickness_world / (Real)thickness_in_particles;
    for (uint32_t k = 0; k < thickness_in_particles; ++k) {
        add_boundary_line(start + offset, end + offset + elongation);
        offset = offset + step;
    }
}

void FluidParticleWorld::add_boundary_line(Point start, Point end) {  // :181-195
    const Vector d = end - start;
    const Real distance = std::sqrt(d.x * d.x + d.y * d.y);
    const Real num_particles_per_meter = properties.num_particles_per_meter();
    const size_t num_shadow_particles = std::max<size_t>(1, (size_t)std::ceil(distance * num_particles_per_meter));
    particles.boundary_particles.reserve(particles.boundary_particles.size() + num_shadow_particles);
    const Vector step = d / distance / num_particles_per_meter;
    Point pos = start;
    for (size_t k = 0; k < num_shadow_particles; ++k) {
        particles.boundary_particles.push_back(pos);
        pos = pos + step;
    }
    boundary_changed = true;
}

void reset_fluid(FluidParticleWorld& w, Real s) {  // main.rs:177-196
    w.remove_all_fluid_particles();
    w.remove_all_boundary_particles();
    w.add_fluid_rect(0.1f * s, 0.7f * s, 0.5f * s, 1.0f * s, 0.05f);
    w.add_boundary_thick_line(Point{0.0f * s, 2.5f * s}, Point{2.0f * s, 2.5f * s}, 4);
    w.add_boundary_thick_line(Point{0.0f * s, 0.0f * s}, Point{2.0f * s, 0.0f * s}, 4);
    w.add_boundary_thick_line(Point{0.0f * s, 0.0f * s}, Point{0.0f * s, 2.5f * s}, 4);
    w.add_boundary_thick_line(Point{2.0f * s, 0.0f * s}, Point{2.0f * s, 2.5f * s}, 4);
    w.add_boundary_thick_line(Point{0.0f * s, 0.6f * s}, Point{1.75f * s, 0.5f * s}, 2);
    w.add_boundary_thick_line(Point{0.0f * s, 2.5f * s}, Point{2.0f * s, 2.5f * s}, 2);
    w.add_boundary_thick_line(Point{-2.0f * s, -0.5f * s}, Point{4.0f * s, -0.5f * s}, 4);
}

// ---- TimeManager ------------------------------------------------------------------------------------------------------
TimeManager TimeManager::adaptive(Duration tmax, Duration tmin, Real cfl) {
    TimeManager t;
    t.fixed = false;
    t.timestep_max = tmax;
    t.timestep_min = tmin;
    t.cfl_factor = cfl;
    t.simulation_step_ = tmin;  // timemanager.rs:106-109
    return t;
}
TimeManager TimeManager::fixed_step(Duration step) {
    TimeManager t;
    t.fixed = true;
    t.timestep_max = t.timestep_min = t.simulation_step_ = step;
    return t;
}
void TimeManager::restart() {
    simulation_step_ = fixed ? timestep_max : timestep_min;
    num_simulation_steps = 0;
    total_simulated_time = Duration{};
}
Duration TimeManager::update_simulation_step(Real particle_diameter, Real max_velocity) {  // timemanager.rs:252-279
    if (!fixed) {
        const Real VELOCITY_EPSILON = 0.00001f;
        const Duration time_cfl = Duration::from_secs_f32(cfl_factor * 0.4f * particle_diameter / (max_velocity + VELOCITY_EPSILON));
        const uint64_t upper_bound = std::min(timestep_max.ns, simulation_step_.mul(2).ns);
        simulation_step_.ns = std::max(lower_bound().ns, std::min(upper_bound, time_cfl.ns));
    }
    return simulation_step_;
}
Duration TimeManager::lower_bound() const {  // timemanager.rs:268-274
    if (timestep_target_frame.ns == 0) return timestep_min;  // AdaptiveTimeStepTarget::None (main.rs:125)
    const uint64_t total = total_simulated_time.ns, target = timestep_target_frame.ns;
    const uint64_t time_to_target = total - target * (uint64_t)(uint32_t)(total / target);  // literally: the remainder
    return Duration{std::min(timestep_min.ns, time_to_target)};
}
void timer_law_of(const TimeManager& tm, Real particle_diameter, sphx_timer_law* out) {
    out->adaptive = tm.fixed ? 0u : 1u;
    out->cfl_factor = tm.fixed ? 1.0f : tm.cfl_factor;
    out->particle_diameter = particle_diameter;
    out->reserved = 0;
    out->timestep_min_ns = tm.fixed ? tm.simulation_step_.ns : tm.lower_bound().ns;
    out->timestep_max_ns = tm.fixed ? tm.simulation_step_.ns : tm.timestep_max.ns;
    out->simulation_step_ns = tm.simulation_step_.ns;
}
void TimeManager::on_step_started() {
    num_simulation_steps += 1;
    total_simulated_time.ns += simulation_step_.ns;
}

// ---- HipDfsphSolver -----------------------------------------------------------------------------------------------------
sphx_params HipDfsphSolver::params_of(const FluidParticleWorld& world, const sphx_params* params) {
    sphx_params p;
    if (params) {
        p = *params;
    } else {
        sphx_default_params(1.0f, 1.0f, 1.0f, &p);
        p.smoothing_length = world.properties.smoothing_length();  // DFSPHSolver::new(xsph, smoothing_length) main.rs:100
        p.particle_mass = world.properties.particle_mass();
        p.fluid_density = world.properties.fluid_density();
        p.particle_radius = world.properties.particle_radius();
        p.gravity[0] = world.gravity.x;
        p.gravity[1] = world.gravity.y;
    }
    return p;
}

HipDfsphSolver::HipDfsphSolver(const FluidParticleWorld& world, const sphx_params* params) {
    const sphx_params p = params_of(world, params);
    last_status = sphx_create(&p, &ctx_);
    if (last_status != SPHX_OK) {
        last_error = sphx_last_error(nullptr);
        ctx_ = nullptr;
    }
}
HipDfsphSolver::~HipDfsphSolver() { sphx_destroy(ctx_); }

void HipDfsphSolver::clear_cached_data() {  // dfsph.rs:406-412
    if (!ctx_) return;
    last_status = sphx_clear_cached(ctx_);
    uploaded_n_ = (size_t)-1;  // next step re-reads the host world, like the reference which always reads it
}

void HipDfsphSolver::simulation_step(FluidParticleWorld& w, TimeManager& tm) {  // dfsph.rs:414-525
    if (!ctx_) {
        last_status = SPHX_ERR_NO_DEVICE;
        return;
    }
    auto fail = [&](int rc) {
        last_status = rc;
        last_error = sphx_last_error(ctx_);
    };
    int rc;
    if (w.boundary_changed) {  // fluidparticleworld.rs:247-252
        const auto& b = w.particles.boundary_particles;
        if ((rc = sphx_set_boundary(ctx_, b.empty() ? nullptr : &b[0].x, (uint32_t)b.size()))) return fail(rc);
        w.boundary_changed = false;
    }
    const size_t n = w.particles.positions.size();
    if (n != uploaded_n_ || w.fluid_generation != uploaded_generation_) {
        // The reference reads the world's Vecs directly; here they are (re)uploaded when the caller edited them.
        if (w.particles.velocities.size() != n) w.particles.velocities.resize(n, Vector{0, 0});
        if (w.stale_prefix) {
            // the caller edited a world whose first stale_prefix particles are behind the device (headless steps): fetch them first,
            // so that e.g. add_fluid_rect mid-run appends to the CURRENT fluid instead of rewinding it to the last sync
            const size_t keep = w.stale_prefix;
            if (keep > n || keep != sphx_num_particles(ctx_)) {
                last_status = SPHX_ERR_NOT_READY;
                last_error = "the host world was edited while it was behind the device state: call sync_world() before editing particles";
                return;
            }
            w.particles.densities.resize(n, 0.0f);
            if ((rc = sphx_download(ctx_, &w.particles.positions[0].x, &w.particles.velocities[0].x, w.particles.densities.data(), nullptr)))
                return fail(rc);
            w.stale_prefix = 0;
        }
        if ((rc = sphx_upload(ctx_, n ? &w.particles.positions[0].x : nullptr, n ? &w.particles.velocities[0].x : nullptr, (uint32_t)n)))
            return fail(rc);
        uploaded_n_ = n;
        uploaded_generation_ = w.fluid_generation;
    }
    if ((rc = device_step(w, tm))) return fail(rc);
    last_status = SPHX_OK;
    if (sync_every_step) {
        if ((rc = sync_world(w))) return fail(rc);
    } else {
        w.stale_prefix = n;  // the host arrays are now behind the device
    }
}

int HipDfsphSolver::device_step(FluidParticleWorld& w, TimeManager& tm) {
    int rc;
    const Real dt_prev = tm.simulation_step().as_secs_f32();  // dfsph.rs:433
    Real vmax = 0;
    // the law the timer is about to apply (public TimerConfig, timemanager.rs:175): the device starts phase B with it right away
    sphx_timer_law law;
    timer_law_of(tm, w.properties.particle_radius() * 2.0f, &law);
    if ((rc = sphx_step_begin_law(ctx_, dt_prev, use_timer_law ? &law : nullptr, &vmax))) return rc;
    const Real dt = tm.update_simulation_step(w.properties.particle_radius() * 2.0f, vmax).as_secs_f32();  // dfsph.rs:478-480
    return sphx_step_finish(ctx_, dt, &last_stats);
}

// ---- HipDfsphMultiSolver ------------------------------------------------------------------------------------------------------
HipDfsphMultiSolver::HipDfsphMultiSolver(const FluidParticleWorld& world, const sphx_params* params, const int* devices, int n_devices,
                                         const sphx_multi_options* options)
    : HipDfsphSolver(NoContext{}) {
    const sphx_params p = params_of(world, params);
    last_status = sphx_multi_create(&p, devices, n_devices, options, &multi_);
    if (last_status != SPHX_OK) {
        last_error = sphx_multi_last_error(nullptr);
        multi_ = nullptr;
    }
}
HipDfsphMultiSolver::~HipDfsphMultiSolver() { sphx_multi_destroy(multi_); }

void HipDfsphMultiSolver::clear_cached_data() {
    if (!multi_) return;
    last_status = sphx_multi_clear_cached(multi_);
    uploaded_n_ = (size_t)-1;
}

void HipDfsphMultiSolver::simulation_step(FluidParticleWorld& w, TimeManager& tm) {
    if (!multi_) {
        last_status = SPHX_ERR_NO_DEVICE;
        return;
    }
    auto fail = [&](int rc) {
        last_status = rc;
        last_error = sphx_multi_last_error(multi_);
    };
    int rc;
    size_t n = w.particles.positions.size();
    if (w.boundary_changed || n != uploaded_n_ || w.fluid_generation != uploaded_generation_) {
        // the tiles are cut from the scene: a changed boundary or particle set means a fresh decomposition
        if (w.stale_prefix) {
            // Headless steps left the first stale_prefix host particles behind the device: fetch them before re-cutting.  What the
            // caller appended behind them since (add_fluid_rect mid-run) is kept: the download replaces the prefix only, like the
            // single-context solver does.
            const size_t keep = w.stale_prefix;
            // (no comparison with the tiles' owned count: particles the tiles have retired — non-finite, out of every band — make it
            // smaller than the host's stale prefix, and in rank mode it is this rank's share only; round-3 advisor finding)
            if (keep > n) {
                last_status = SPHX_ERR_NOT_READY;
                last_error = "the host world was edited while it was behind the device state: call sync_world() before editing particles";
                return;
            }
            if (w.particles.velocities.size() != n) w.particles.velocities.resize(n, Vector{0, 0});
            const std::vector<Point> tail_p(w.particles.positions.begin() + keep, w.particles.positions.end());
            const std::vector<Vector> tail_v(w.particles.velocities.begin() + keep, w.particles.velocities.end());
            if ((rc = sync_world(w))) return fail(rc);  // (resizes the arrays to the owned particles)
            w.particles.positions.insert(w.particles.positions.end(), tail_p.begin(), tail_p.end());
            w.particles.velocities.insert(w.particles.velocities.end(), tail_v.begin(), tail_v.end());
            w.particles.densities.resize(w.particles.positions.size(), 0.0f);
            w.particles.particle_ids.clear();  // the upload below numbers the particles afresh
            n = w.particles.positions.size();
        }
        const auto& b = w.particles.boundary_particles;
        if ((rc = sphx_multi_set_boundary(multi_, b.empty() ? nullptr : &b[0].x, (uint32_t)b.size()))) return fail(rc);
        w.boundary_changed = false;
        if (w.particles.velocities.size() != n) w.particles.velocities.resize(n, Vector{0, 0});
        if ((rc = sphx_multi_upload(multi_, n ? &w.particles.positions[0].x : nullptr, n ? &w.particles.velocities[0].x : nullptr, nullptr, (uint32_t)n)))
            return fail(rc);
        uploaded_n_ = n;
        uploaded_generation_ = w.fluid_generation;
    }
    const Real dt_prev = tm.simulation_step().as_secs_f32();  // dfsph.rs:433
    Real vmax = 0;
    if ((rc = sphx_multi_step_begin(multi_, dt_prev, &vmax))) return fail(rc);
    const Real dt = tm.update_simulation_step(w.properties.particle_radius() * 2.0f, vmax).as_secs_f32();  // dfsph.rs:478-480
    if ((rc = sphx_multi_step_finish(multi_, dt, &last_stats))) return fail(rc);
    last_status = SPHX_OK;
    if (sync_every_step) {
        if ((rc = sync_world(w))) return fail(rc);
    } else {
        w.stale_prefix = n;
    }
}

int HipDfsphMultiSolver::sync_world(FluidParticleWorld& w) {
    if (!multi_) return SPHX_ERR_NO_DEVICE;
    uint64_t n = sphx_multi_num_owned(multi_);
    w.particles.positions.resize(n);
    w.particles.velocities.resize(n);
    w.particles.densities.resize(n);
    w.particles.particle_ids.resize(n);
    const int rc = sphx_multi_download(multi_, n ? &w.particles.positions[0].x : nullptr, n ? &w.particles.velocities[0].x : nullptr,
                                       n ? w.particles.densities.data() : nullptr, n ? w.particles.particle_ids.data() : nullptr, &n);
    w.stale_prefix = 0;
    uploaded_generation_ = w.fluid_generation;
    uploaded_n_ = n;
    return rc;
}

// WCSPHSolver<XSPHViscosityModel>::simulation_step, wscsph.rs:126-179
int HipWcsphSolver::device_step(FluidParticleWorld& w, TimeManager& tm) {
    int rc;
    const Real dt = tm.simulation_step().as_secs_f32();  // wscsph.rs:135
    Real vmax = 0;
    if ((rc = sphx_wcsph_step_begin(ctx_, dt, &vmax))) return rc;
    const Real dt_new = tm.update_simulation_step(w.properties.particle_radius() * 2.0f, vmax).as_secs_f32();  // wscsph.rs:162-164
    return sphx_wcsph_step_finish(ctx_, dt_new, &last_stats);
}

int HipDfsphSolver::sync_world(FluidParticleWorld& w) {
    if (!ctx_) return SPHX_ERR_NO_DEVICE;
    const size_t n = sphx_num_particles(ctx_);
    w.particles.positions.resize(n);
    w.particles.velocities.resize(n);
    w.particles.densities.resize(n);
    w.particles.particle_ids.resize(n);
    int rc = sphx_download(ctx_, n ? &w.particles.positions[0].x : nullptr, n ? &w.particles.velocities[0].x : nullptr,
                           n ? w.particles.densities.data() : nullptr, n ? w.particles.particle_ids.data() : nullptr);
    // host arrays now equal the device state: no re-upload needed
    w.stale_prefix = 0;
    uploaded_generation_ = w.fluid_generation;
    uploaded_n_ = n;
    return rc;
}

}  // namespace sph

// =====================================================================================================================
// C exports (include/sphx.h, host-side mirror section)
// =====================================================================================================================
struct sphx_world {
    sph::FluidParticleWorld w;
    sphx_world(float a, float b, float c) : w(a, b, c) {}
};
struct sphx_timer {
    sph::TimeManager t;
};
struct sphx_solver {
    sph::HipDfsphSolver* sp;
    sph::HipDfsphSolver& s;
    sphx_solver(sph::HipDfsphSolver* p) : sp(p), s(*p) {}
    ~sphx_solver() { delete sp; }
};

extern "C" {

sphx_world* sphx_world_create(float smoothing_factor, float particle_density, float fluid_density) {
    if (!(particle_density > 0)) return nullptr;
    return new sphx_world(smoothing_factor, particle_density, fluid_density);
}
void sphx_world_destroy(sphx_world* w) { delete w; }
void sphx_world_properties(const sphx_world* w, float* out4) {
    out4[0] = w->w.properties.smoothing_length();
    out4[1] = w->w.properties.particle_mass();
    out4[2] = w->w.properties.particle_radius();
    out4[3] = w->w.properties.fluid_density();
}
void sphx_world_remove_all_fluid_particles(sphx_world* w) { w->w.remove_all_fluid_particles(); }
void sphx_world_remove_all_boundary_particles(sphx_world* w) { w->w.remove_all_boundary_particles(); }
void sphx_world_add_fluid_rect(sphx_world* w, float x, float y, float width, float height, float jitter) {
    w->w.add_fluid_rect(x, y, width, height, jitter);
}
void sphx_world_add_boundary_thick_line(sphx_world* w, float sx, float sy, float ex, float ey, uint32_t thickness) {
    w->w.add_boundary_thick_line(sph::Point{sx, sy}, sph::Point{ex, ey}, thickness);
}
void sphx_world_add_boundary_line(sphx_world* w, float sx, float sy, float ex, float ey) {
    w->w.add_boundary_line(sph::Point{sx, sy}, sph::Point{ex, ey});
}
void sphx_world_reset_fluid(sphx_world* w, float scale) { sph::reset_fluid(w->w, scale); }
uint32_t sphx_world_num_dynamic_particles(const sphx_world* w) { return (uint32_t)w->w.particles.num_dynamic_particles(); }
uint32_t sphx_world_num_boundary_particles(const sphx_world* w) { return (uint32_t)w->w.particles.num_boundary_particles(); }
float* sphx_world_positions(sphx_world* w) { return w->w.particles.positions.empty() ? nullptr : &w->w.particles.positions[0].x; }
float* sphx_world_velocities(sphx_world* w) { return w->w.particles.velocities.empty() ? nullptr : &w->w.particles.velocities[0].x; }
float* sphx_world_densities(sphx_world* w) { return w->w.particles.densities.empty() ? nullptr : w->w.particles.densities.data(); }
float* sphx_world_boundary(sphx_world* w) {
    return w->w.particles.boundary_particles.empty() ? nullptr : &w->w.particles.boundary_particles[0].x;
}
uint32_t* sphx_world_particle_ids(sphx_world* w) { return w->w.particles.particle_ids.empty() ? nullptr : w->w.particles.particle_ids.data(); }
void sphx_world_set_particles(sphx_world* w, const float* pos_xy, const float* vel_xy, uint32_t n) {
    auto& p = w->w.particles;
    p.positions.resize(n);
    p.velocities.resize(n);
    p.densities.assign(n, 0.0f);
    for (uint32_t i = 0; i < n; ++i) {
        p.positions[i] = sph::Point{pos_xy[2 * i], pos_xy[2 * i + 1]};
        p.velocities[i] = vel_xy ? sph::Vector{vel_xy[2 * i], vel_xy[2 * i + 1]} : sph::Vector{0, 0};
    }
    w->w.stale_prefix = 0;  // everything replaced
    w->w.fluid_generation++;
}
void sphx_world_set_boundary(sphx_world* w, const float* xy, uint32_t n) {
    auto& b = w->w.particles.boundary_particles;
    b.resize(n);
    for (uint32_t i = 0; i < n; ++i) b[i] = sph::Point{xy[2 * i], xy[2 * i + 1]};
    w->w.boundary_changed = true;
}
void sphx_world_set_gravity(sphx_world* w, float gx, float gy) { w->w.gravity = sph::Vector{gx, gy}; }

uint64_t sphx_duration_from_secs_f32(float secs) { return sph::Duration::from_secs_f32(secs).ns; }
float sphx_duration_as_secs_f32(uint64_t nanos) { return sph::Duration{nanos}.as_secs_f32(); }
sphx_timer* sphx_timer_create_adaptive(uint64_t tmax, uint64_t tmin, float cfl) {
    return new sphx_timer{sph::TimeManager::adaptive(sph::Duration{tmax}, sph::Duration{tmin}, cfl)};
}
sphx_timer* sphx_timer_create_fixed(uint64_t step) { return new sphx_timer{sph::TimeManager::fixed_step(sph::Duration{step})}; }
void sphx_timer_destroy(sphx_timer* t) { delete t; }
void sphx_timer_restart(sphx_timer* t) { t->t.restart(); }
uint64_t sphx_timer_simulation_step_ns(const sphx_timer* t) { return t->t.simulation_step().ns; }
uint64_t sphx_timer_update_simulation_step(sphx_timer* t, float diameter, float vmax) { return t->t.update_simulation_step(diameter, vmax).ns; }
uint64_t sphx_timer_total_simulated_ns(const sphx_timer* t) { return t->t.total_simulated_time.ns; }
uint32_t sphx_timer_num_steps(const sphx_timer* t) { return t->t.num_simulation_steps; }
void sphx_timer_set_target_frame(sphx_timer* t, uint64_t target_ns) { t->t.timestep_target_frame.ns = target_ns; }
void sphx_timer_on_step_started(sphx_timer* t) { t->t.on_step_started(); }
int sphx_timer_law_of(const sphx_timer* t, float particle_diameter, sphx_timer_law* out) {
    if (!t || !out) return SPHX_ERR_INVALID_ARGUMENT;
    sph::timer_law_of(t->t, particle_diameter, out);
    return SPHX_OK;
}

int sphx_solver_create_dfsph(const sphx_world* w, const sphx_params* params, sphx_solver** out) {
    if (!w || !out) return SPHX_ERR_INVALID_ARGUMENT;
    sphx_solver* s = new sphx_solver(new sph::HipDfsphSolver(w->w, params));
    if (!s->s.ok()) {
        const int rc = s->s.last_status;
        delete s;
        *out = nullptr;
        return rc;
    }
    *out = s;
    return SPHX_OK;
}
int sphx_solver_create_dfsph_multi(const sphx_world* w, const sphx_params* params, const int* devices, int n_devices, const sphx_multi_options* options,
                                   sphx_solver** out) {
    if (!w || !out || !devices || n_devices < 1) return SPHX_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    sphx_solver* s = new sphx_solver(new sph::HipDfsphMultiSolver(w->w, params, devices, n_devices, options));
    if (!s->s.ok()) {
        const int rc = s->s.last_status;
        delete s;
        return rc;
    }
    *out = s;
    return SPHX_OK;
}
int sphx_solver_create_wcsph(const sphx_world* w, const sphx_params* params, sphx_solver** out) {
    if (!w || !out) return SPHX_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    sphx_solver* s = new sphx_solver(new sph::HipWcsphSolver(w->w, params));
    if (!s->s.ok()) {
        const int rc = s->s.last_status;
        delete s;
        return rc;
    }
    *out = s;
    return SPHX_OK;
}
void sphx_solver_destroy(sphx_solver* s) { delete s; }
void sphx_solver_clear_cached_data(sphx_solver* s) { s->s.clear_cached_data(); }
int sphx_solver_simulation_step(sphx_solver* s, sphx_world* w, sphx_timer* t, int sync_world, sphx_step_stats* out) {
    if (!s || !w || !t) return SPHX_ERR_INVALID_ARGUMENT;
    s->s.sync_every_step = sync_world != 0;
    t->t.on_step_started();
    s->s.simulation_step(w->w, t->t);
    if (out) *out = s->s.last_stats;
    return s->s.last_status;
}
int sphx_solver_simulation_steps(sphx_solver* s, sphx_world* w, sphx_timer* t, int sync_world, uint32_t k, sphx_step_stats* out, uint32_t* out_done) {
    if (out_done) *out_done = 0;
    if (!s || !w || !t) return SPHX_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < k; ++i) {
        const int rc = sphx_solver_simulation_step(s, w, t, sync_world, out ? out + i : nullptr);
        if (rc) return rc;
        if (out_done) *out_done = i + 1;
    }
    return SPHX_OK;
}
int sphx_solver_sync_world(sphx_solver* s, sphx_world* w) { return s->s.sync_world(w->w); }
sphx_ctx* sphx_solver_ctx(sphx_solver* s) { return s->s.ctx(); }
const char* sphx_solver_last_error(const sphx_solver* s) { return s->s.last_error.c_str(); }

}  // extern "C"

// =====================================================================================================================
// single-node scalar all-reduce through POSIX shared memory (tile driver)
// =====================================================================================================================
namespace {
constexpr int SHM_MAX_WORLD = 64;
constexpr int SHM_MAX_N = 8;
struct ShmSegment {
    std::atomic<uint32_t> magic;
    std::atomic<uint32_t> attached;
    std::atomic<uint64_t> arrive;  // total arrivals over all epochs
    // a rank that fails (or closes) raises this; the ranks spinning in an all-reduce return at once instead of waiting for the time-out
    std::atomic<uint32_t> abort;
    std::atomic<uint32_t> go;  // rank 0: every rank of THIS run has joined THIS segment
    // join handshake: rank r publishes a random token, rank 0 echoes it.  A segment a crashed earlier run left under the same name
    // has nobody echoing: a rank that got there before rank 0 replaced it notices and attaches again.
    std::atomic<uint64_t> token[SHM_MAX_WORLD], echo[SHM_MAX_WORLD];
    double slots[2][SHM_MAX_WORLD][SHM_MAX_N];
};
double shm_timeout_s() {  // SPHX_SHM_TIMEOUT_S: how long an all-reduce waits for a rank that never arrives (default 300 s; tests: 2)
    if (const char* e = std::getenv("SPHX_SHM_TIMEOUT_S")) {
        const double v = std::atof(e);
        if (v > 0) return v;
    }
    return 300.0;
}
uint64_t shm_random_token() {
    uint64_t t = 0;
    if (FILE* f = std::fopen("/dev/urandom", "rb")) {
        if (std::fread(&t, sizeof(t), 1, f) != 1) t = 0;
        std::fclose(f);
    }
    t ^= (uint64_t)getpid() << 32 ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    return t ? t : 1;
}
}  // namespace
struct sphx_shm {
    ShmSegment* seg = nullptr;
    int rank = 0, world = 1;
    uint64_t epoch = 0;
    double timeout_s = 300.0;
    std::string name;
};

extern "C" {

sphx_shm* sphx_shm_open(const char* name, int rank, int world) {
    if (!name || rank < 0 || world < 1 || rank >= world || world > SHM_MAX_WORLD) return nullptr;
    const std::string nm = std::string("/sphx_") + name;
    const double join_timeout = std::min(shm_timeout_s(), 120.0);
    const auto t0 = std::chrono::steady_clock::now();
    auto late = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > join_timeout; };
    auto map = [&](int fd) -> ShmSegment* {
        void* p = mmap(nullptr, sizeof(ShmSegment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        return p == MAP_FAILED ? nullptr : (ShmSegment*)p;
    };
    sphx_shm* h = new sphx_shm();
    h->rank = rank;
    h->world = world;
    h->name = nm;
    h->timeout_s = shm_timeout_s();
    if (rank == 0) {
        shm_unlink(nm.c_str());  // whatever an earlier run left under this name is not ours
        const int fd = shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(ShmSegment)) != 0) {  // (a fresh segment is zero-filled: no tokens, no arrivals)
            if (fd >= 0) close(fd);
            delete h;
            return nullptr;
        }
        h->seg = map(fd);
        close(fd);
        if (!h->seg) {
            delete h;
            return nullptr;
        }
        h->seg->magic.store(0x53504858u, std::memory_order_release);
        // echo every rank's token as it appears
        int joined = 1;
        std::vector<uint8_t> seen(world, 0);
        while (joined < world) {
            for (int r = 1; r < world; ++r) {
                const uint64_t t = h->seg->token[r].load(std::memory_order_acquire);
                if (t && !seen[r]) {
                    h->seg->echo[r].store(t, std::memory_order_release);
                    seen[r] = 1;
                    joined += 1;
                }
            }
            if (joined < world) {
                if (late()) {
                    munmap(h->seg, sizeof(ShmSegment));
                    shm_unlink(nm.c_str());
                    delete h;
                    return nullptr;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        }
        h->seg->attached.store((uint32_t)world);
        h->seg->go.store(1, std::memory_order_release);
        return h;
    }
    const uint64_t my_token = shm_random_token();
    for (;;) {  // attach; if nobody echoes the token the segment is a stale one: attach again
        int fd = -1;
        while ((fd = shm_open(nm.c_str(), O_RDWR, 0600)) < 0) {
            if (late()) {
                delete h;
                return nullptr;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        struct stat st;
        if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(ShmSegment)) {
            // rank 0 has created the file but not sized it yet — or this is a leftover of another layout: look the name up again
            close(fd);
            if (late()) {
                delete h;
                return nullptr;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            continue;
        }
        const ino_t ino = st.st_ino;
        ShmSegment* seg = map(fd);
        close(fd);
        if (!seg) {
            delete h;
            return nullptr;
        }
        bool stale = false;
        while (seg->magic.load(std::memory_order_acquire) != 0x53504858u && !late()) std::this_thread::yield();
        seg->token[rank].store(my_token, std::memory_order_release);
        auto t_check = std::chrono::steady_clock::now();
        while (!(seg->echo[rank].load(std::memory_order_acquire) == my_token && seg->go.load(std::memory_order_acquire))) {
            if (late()) {
                munmap(seg, sizeof(ShmSegment));
                delete h;
                return nullptr;
            }
            if (std::chrono::steady_clock::now() - t_check > std::chrono::milliseconds(20)) {
                // does the name still lead to the segment that is mapped here?
                t_check = std::chrono::steady_clock::now();
                const int fd2 = shm_open(nm.c_str(), O_RDWR, 0600);
                struct stat st2;
                const bool same = fd2 >= 0 && fstat(fd2, &st2) == 0 && st2.st_ino == ino;
                if (fd2 >= 0) close(fd2);
                if (!same) {
                    stale = true;
                    break;
                }
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        if (stale) {
            munmap(seg, sizeof(ShmSegment));
            continue;
        }
        h->seg = seg;
        return h;
    }
}

void sphx_shm_abort(sphx_shm* h) {
    if (h && h->seg) h->seg->abort.store(1, std::memory_order_release);
}

// one meeting of all ranks: this rank's n <= 8 doubles go into its slot; returns once every rank's are there (buffer index in *buf)
static int shm_meet(sphx_shm* h, const double* in, int n, int* buf_out) {
    ShmSegment* s = h->seg;
    const int buf = (int)(h->epoch & 1);
    for (int k = 0; k < n; ++k) s->slots[buf][h->rank][k] = in[k];
    h->epoch += 1;
    const uint64_t target = h->epoch * (uint64_t)h->world;
    s->arrive.fetch_add(1, std::memory_order_acq_rel);
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    while (s->arrive.load(std::memory_order_acquire) < target) {
        if (s->abort.load(std::memory_order_acquire)) {
            // (a rank that has finished raises the word when it closes: everybody had arrived by then — look again before giving up)
            if (s->arrive.load(std::memory_order_acquire) >= target) break;
            return SPHX_ERR_NOT_READY;
        }
        if ((++spins & 0x3FFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > h->timeout_s) {
            s->abort.store(1, std::memory_order_release);  // nobody else needs to sit out the same time-out
            return SPHX_ERR_NOT_READY;
        }
        __builtin_ia32_pause();
    }
    *buf_out = buf;
    return SPHX_OK;
}

int sphx_shm_allreduce(sphx_shm* h, const double* in, int n, int op, double* out) {
    if (!h || !in || !out || n < 1 || n > SHM_MAX_N) return SPHX_ERR_INVALID_ARGUMENT;
    int buf;
    const int rc = shm_meet(h, in, n, &buf);
    if (rc) return rc;
    ShmSegment* s = h->seg;
    for (int k = 0; k < n; ++k) {
        double acc = s->slots[buf][0][k];
        for (int r = 1; r < h->world; ++r) {
            const double v = s->slots[buf][r][k];
            acc = op == 1 ? (v > acc ? v : acc) : acc + v;
        }
        out[k] = acc;
    }
    return SPHX_OK;
}

// every rank's n doubles to every rank: out[r * n + k] = rank r's in[k] (the halo exchange's per-peer record counts travel this way)
int sphx_shm_allgather(sphx_shm* h, const double* in, int n, double* out) {
    if (!h || !in || !out || n < 1 || n > SHM_MAX_N) return SPHX_ERR_INVALID_ARGUMENT;
    int buf;
    const int rc = shm_meet(h, in, n, &buf);
    if (rc) return rc;
    ShmSegment* s = h->seg;
    for (int r = 0; r < h->world; ++r)
        for (int k = 0; k < n; ++k) out[r * n + k] = s->slots[buf][r][k];
    return SPHX_OK;
}

void sphx_shm_close(sphx_shm* h) {
    if (!h) return;
    if (h->seg) {
        h->seg->abort.store(1, std::memory_order_release);  // whoever still waits for this rank will not get it
        munmap(h->seg, sizeof(ShmSegment));
        if (h->rank == 0) shm_unlink(h->name.c_str());
    }
    delete h;
}

}  // extern "C"
