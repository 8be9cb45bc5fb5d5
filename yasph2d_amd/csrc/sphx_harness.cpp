// sphx_harness — headless driver over the C++ host mirror (sphx_host.hpp): the part of the reference app's loop that does not
// draw (main.rs:85-129 set-up, :177-196 scene, :279 `sph_solver.simulation_step(&mut fluid_world, &mut time_manager)`).
//
//   sphx_harness [--solver dfsph|wcsph] [--scale S | --particles N] [--steps K] [--warmup W] [--no-law] [--sync]
//
// Prints one JSON line: particle-steps/s over the K timed steps, the timer's final step, iteration statistics and an FNV-1a
// checksum of the final (downloaded) positions/velocities, which tests compare with the Python-driven run of the same scene.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "sphx_host.hpp"

static uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}

int main(int argc, char** argv) {
    std::string solver_kind = "dfsph";
    float scale = 1.0f;
    long steps = 100, warmup = 5;
    bool law = true, sync = false;
    for (int a = 1; a < argc; ++a) {
        const std::string s = argv[a];
        auto next = [&]() -> const char* { return a + 1 < argc ? argv[++a] : "0"; };
        if (s == "--solver") solver_kind = next();
        else if (s == "--scale") scale = (float)std::atof(next());
        else if (s == "--particles") scale = (float)std::sqrt(std::atof(next()) / 4050.0);
        else if (s == "--steps") steps = std::atol(next());
        else if (s == "--warmup") warmup = std::atol(next());
        else if (s == "--no-law") law = false;
        else if (s == "--sync") sync = true;
        else {
            std::fprintf(stderr, "unknown argument %s\n", s.c_str());
            return 2;
        }
    }
    const bool wcsph = solver_kind == "wcsph";
    sph::FluidParticleWorld world(2.0f, 10000.0f, 100.0f);  // main.rs:85-89
    sph::reset_fluid(world, scale);                          // main.rs:177-196
    std::unique_ptr<sph::HipDfsphSolver> solver(wcsph ? new sph::HipWcsphSolver(world, nullptr) : new sph::HipDfsphSolver(world, nullptr));
    if (!solver->ok()) {
        std::fprintf(stderr, "solver: %s (status %d)\n", solver->last_error.c_str(), solver->last_status);
        return 1;  // no CPU fallback
    }
    solver->sync_every_step = sync;
    solver->use_timer_law = law;
    sph::TimeManager tm = sph::TimeManager::adaptive(sph::Duration::from_secs_f32(1.0f / 120.0f / 3.0f), sph::Duration::from_secs_f32(1.0f / 60.0f / 400.0f),
                                                     wcsph ? 0.2f : 1.5f);  // main.rs:115-127
    const size_t n = world.particles.positions.size();
    auto step = [&]() {
        tm.on_step_started();
        solver->simulation_step(world, tm);
        if (solver->last_status != SPHX_OK) {
            std::fprintf(stderr, "step failed: %s (status %d)\n", solver->last_error.c_str(), solver->last_status);
            std::exit(1);
        }
    };
    for (long i = 0; i < warmup; ++i) step();
    sphx_synchronize(solver->ctx());
    unsigned long long id_sum = 0, iv_sum = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (long i = 0; i < steps; ++i) {
        step();
        id_sum += solver->last_stats.density_iterations;
        iv_sum += solver->last_stats.divergence_iterations;
    }
    sphx_synchronize(solver->ctx());
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (solver->sync_world(world) != SPHX_OK) return 1;
    // order-independent content check: positions/velocities placed at their persistent particle id
    std::vector<float> by_id(4 * n, 0.0f);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t id = world.particles.particle_ids[i];
        by_id[4 * id + 0] = world.particles.positions[i].x;
        by_id[4 * id + 1] = world.particles.positions[i].y;
        by_id[4 * id + 2] = world.particles.velocities[i].x;
        by_id[4 * id + 3] = world.particles.velocities[i].y;
    }
    std::printf("{\"solver\": \"%s\", \"particles\": %zu, \"boundary\": %zu, \"steps\": %ld, \"particle_steps_per_s\": %.6e, \"ms_per_step\": %.6f, "
                "\"timer_step_ns\": %llu, \"simulated_ns\": %llu, \"mean_density_iterations\": %.4f, \"mean_divergence_iterations\": %.4f, "
                "\"state_fnv1a\": \"%016llx\"}\n",
                solver_kind.c_str(), n, world.particles.boundary_particles.size(), steps, (double)n * (double)steps / el, el / (double)steps * 1e3,
                (unsigned long long)tm.simulation_step().ns, (unsigned long long)tm.total_simulated_time.ns, steps ? (double)id_sum / steps : 0.0,
                steps ? (double)iv_sum / steps : 0.0, (unsigned long long)fnv1a(by_id.data(), by_id.size() * 4));
    return 0;
}
