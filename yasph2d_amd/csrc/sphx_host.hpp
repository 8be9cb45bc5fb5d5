// sphx_host.hpp — C++ host-side mirror of the reference's caller-facing types for the DFSPH hot path.
//
// The reference host is Rust; this image has no Rust toolchain, so the host side above the C ABI is written in C++ with
// the reference's names, argument meaning and error behaviour (see INTEGRATION.md for the Rust shim a maintainer would
// add instead).  Only what the hot path and its drivers need is mirrored:
//   sph::ConstantFluidProperties, sph::FluidParticleWorld   src/sph/fluidparticleworld.rs
//   sph::Duration, sph::TimeManager (simulation-step part)  src/sph/timemanager.rs
//   sph::Solver, sph::HipDfsphSolver                        src/sph/solver/{mod,dfsph}.rs
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sphx.h"

namespace sph {

using Real = float;  // units.rs:2
struct Point {       // cgmath::Point2<f32>, units.rs:3
    Real x, y;
};
struct Vector {  // cgmath::Vector2<f32>, units.rs:4
    Real x, y;
};

// std::time::Duration restricted to what timemanager.rs needs (sub-second arithmetic on nanoseconds)
struct Duration {
    uint64_t ns = 0;
    static Duration from_secs_f32(Real secs);  // round-to-nearest-even nanoseconds (Rust >= 1.63)
    Real as_secs_f32() const;                  // (secs as f32) + (nanos as f32) / 1e9
    Duration mul(uint32_t k) const { return Duration{ns * k}; }
};

// fluidparticleworld.rs:46-90
struct ConstantFluidProperties {
    Real smoothing_length_, particle_density_, fluid_density_;
    ConstantFluidProperties(Real smoothing_factor, Real particle_density, Real fluid_density);
    Real smoothing_length() const { return smoothing_length_; }
    Real fluid_density() const { return fluid_density_; }
    Real particle_mass() const { return fluid_density_ / particle_density_; }
    Real num_particles_per_meter() const;
    static Real particle_radius_from_particle_density(Real particle_density);
    Real particle_radius() const { return particle_radius_from_particle_density(particle_density_); }
};

// fluidparticleworld.rs:11-23 (the neighbourhood structure lives on the device)
struct Particles {
    std::vector<Point> positions;
    std::vector<Vector> velocities;
    std::vector<Real> densities;
    std::vector<Point> boundary_particles;
    std::vector<uint32_t> particle_ids;  // mirror extra: id of each slot after a synced step
    size_t num_dynamic_particles() const { return positions.size(); }
    size_t num_boundary_particles() const { return boundary_particles.size(); }
};

// fluidparticleworld.rs:92-195
struct FluidParticleWorld {
    Particles particles;
    ConstantFluidProperties properties;
    Vector gravity;
    bool boundary_changed;
    uint64_t fluid_generation = 1;  // bumped whenever the host arrays are edited by the caller
    // Headless stepping (sync_world = 0) leaves the first `stale_prefix` host particles behind the device state.  The reference's
    // caller always sees a current world (the solver works on its Vecs in place), so an edit that KEEPS those particles — appending
    // with add_fluid_rect — must not rewind them: the solver downloads the prefix before it uploads the edited arrays.  Edits that
    // replace everything (remove_all_fluid_particles, sphx_world_set_particles) reset it.
    size_t stale_prefix = 0;

    FluidParticleWorld(Real smoothing_factor, Real particle_density, Real fluid_density);
    void remove_all_fluid_particles();
    void remove_all_boundary_particles();
    void add_fluid_rect(Real x, Real y, Real w, Real h, Real jitter_amount);
    void add_boundary_thick_line(Point start, Point end, uint32_t thickness_in_particles);
    void add_boundary_line(Point start, Point end);
};

// main.rs:177-196 with every coordinate multiplied by `scale`
void reset_fluid(FluidParticleWorld& world, Real scale);

// timemanager.rs (simulation clock only; wall/render clocks and frame pacing are viewer concerns)
struct TimeManager {
    bool fixed = false;
    Duration timestep_max, timestep_min;
    Real cfl_factor = 0;
    Duration simulation_step_;
    Duration timestep_target_frame;  // AdaptiveTimeStepTarget::TargetFrameLength (timemanager.rs:24-36); zero = ::None
    uint32_t num_simulation_steps = 0;
    Duration total_simulated_time;

    static TimeManager adaptive(Duration timestep_max, Duration timestep_min, Real cfl_factor);
    static TimeManager fixed_step(Duration step);
    void restart();                                                               // :131-133
    Duration simulation_step() const { return simulation_step_; }                 // :136-138
    Duration update_simulation_step(Real particle_diameter, Real max_velocity);   // :252-279
    Duration lower_bound() const;                                                 // :268-274, known before the update
    void on_step_started();  // the clock part of simulation_frame_loop (:244-247)
};

// what sphx_step_begin_law needs to know about the timer (its public config + current step)
void timer_law_of(const TimeManager& tm, Real particle_diameter, sphx_timer_law* out);

// solver/mod.rs:12-18
struct Solver {
    virtual ~Solver() {}
    virtual void clear_cached_data() = 0;
    virtual void simulation_step(FluidParticleWorld& fluid_world, TimeManager& time_manager) = 0;
};

// DFSPHSolver<XSPHViscosityModel> (dfsph.rs:16-61) running on the HIP device through the C ABI of sphx.h
class HipDfsphSolver : public Solver {
   public:
    HipDfsphSolver(const FluidParticleWorld& world, const sphx_params* params_or_null);
    ~HipDfsphSolver() override;
    virtual bool ok() const { return ctx_ != nullptr; }
    void clear_cached_data() override;
    void simulation_step(FluidParticleWorld& fluid_world, TimeManager& time_manager) override;
    virtual int sync_world(FluidParticleWorld& fluid_world);  // download positions/velocities/densities into the host world

    bool sync_every_step = true;   // main.rs draws from the host arrays after each step
    bool use_timer_law = true;     // sphx_step_begin_law: the device derives dt itself, the host only verifies it
    int last_status = SPHX_OK;     // the trait returns (); failures (reference: panics) are reported here
    std::string last_error;
    sphx_step_stats last_stats{};
    sphx_ctx* ctx() { return ctx_; }

   protected:
    struct NoContext {};
    explicit HipDfsphSolver(NoContext) {}  // for solvers that hold their device state elsewhere (HipDfsphMultiSolver)
    static sphx_params params_of(const FluidParticleWorld& world, const sphx_params* params_or_null);
    virtual int device_step(FluidParticleWorld& fluid_world, TimeManager& time_manager);  // the two-phase step on the device
    sphx_ctx* ctx_ = nullptr;

   private:
    uint64_t uploaded_generation_ = 0;
    size_t uploaded_n_ = (size_t)-1;
};

// The same Solver over SEVERAL GPUs: main.rs keeps its one Box<dyn Solver> and its one call per step (main.rs:50, :279); the domain
// decomposition, the halo exchange and the reductions happen inside libsphx (sphx_multi_*, csrc/sphx_tiles.cpp).  `devices`: one
// HIP device ordinal per tile (an ordinal may repeat).
class HipDfsphMultiSolver : public HipDfsphSolver {
   public:
    HipDfsphMultiSolver(const FluidParticleWorld& world, const sphx_params* params_or_null, const int* devices, int n_devices,
                        const sphx_multi_options* options_or_null);
    ~HipDfsphMultiSolver() override;
    bool ok() const override { return multi_ != nullptr; }
    void clear_cached_data() override;
    void simulation_step(FluidParticleWorld& fluid_world, TimeManager& time_manager) override;
    int sync_world(FluidParticleWorld& fluid_world) override;
    sphx_multi* multi() { return multi_; }

   private:
    sphx_multi* multi_ = nullptr;
    uint64_t uploaded_generation_ = 0;
    size_t uploaded_n_ = (size_t)-1;
};

// WCSPHSolver<XSPHViscosityModel> (wscsph.rs:14-49) on the same device context type: same world handling, the WCSPH step
class HipWcsphSolver : public HipDfsphSolver {
   public:
    using HipDfsphSolver::HipDfsphSolver;

   protected:
    int device_step(FluidParticleWorld& fluid_world, TimeManager& time_manager) override;
};

}  // namespace sph
