"""WCSPH on the device (solver/wscsph.rs) against the oracle's restatement: leap frog, Poly6 densities,
Tait pressure + Spiky gradient + XSPH + boundary force, CFL timer with the app's WCSPH factor 0.2 (main.rs:116-119).
Bar: everything bit-identical, every step."""
import numpy as np
import pytest
from util import assert_bits_equal, assert_same_neighbors, bench_world, dam_break

import yasph2d_amd as y
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu

DIAM = np.float32(0.01)


def wcsph_timer():
    return y.TimeManager(cfl_factor=0.2)


def pair(pos, boundary):
    ctx = y.SphxContext()
    o = Oracle()
    t = wcsph_timer()
    o.timer_adaptive(t.timestep_max_ns, t.timestep_min_ns, 0.2)
    ctx.set_boundary(boundary)
    o.set_boundary(boundary)
    ctx.upload(pos)
    o.set_particles(pos)
    return ctx, o, t


def step_both(ctx, o, timer):
    vmax = ctx.wcsph_step_begin(timer.simulation_step())
    dt_ns = timer.update_simulation_step(DIAM, vmax)
    st = ctx.wcsph_step_finish(y.duration_as_secs_f32(dt_ns))
    so = o.wcsph_step()
    assert dt_ns == o.timer_step_ns()
    assert np.float32(vmax) == np.float32(so["vmax"])
    assert st["neighbor_entries"] == so["neighbor_entries"]
    return st


def compare(ctx, o, what):
    d = ctx.download()
    np.testing.assert_array_equal(d["ids"], o.ids())
    assert_bits_equal(d["pos"], o.positions(), what + " positions")
    assert_bits_equal(d["vel"], o.velocities(), what + " velocities")
    assert_bits_equal(d["density"], o.densities(), what + " densities")


def test_wcsph_dam_break_600_steps():
    """The reference scene through free fall and the impact on the floor (pressure clamp active, boundary forces, dt shrinking
    under the CFL law)."""
    pos, boundary = dam_break(1.0)
    ctx, o, timer = pair(pos, boundary)
    dts = set()
    for s in range(600):
        st = step_both(ctx, o, timer)
        dts.add(st["dt"])
        if s % 100 == 99:
            compare(ctx, o, f"step {s}")
    assert len(dts) > 5, "the adaptive timer must have moved"
    assert_same_neighbors(ctx.download_neighbors(), o.neighbors())


def test_wcsph_bench_world_config0():
    """BASELINE configs[0]: the 8 100 + 4 420 bench world (benches/benchmarks/update_densities.rs:72-80), 100 WCSPH steps."""
    pos, boundary = bench_world()
    ctx, o, timer = pair(pos, boundary)
    for s in range(100):
        step_both(ctx, o, timer)
    compare(ctx, o, "final")


def test_wcsph_clear_cached_and_growth():
    """clear_cached_data drops the accelerations (wscsph.rs:122-124); adding particles keeps the old slots' accelerations
    (Vec::resize, wscsph.rs:129)."""
    pos, boundary = dam_break(1.0)
    ctx, o, timer = pair(pos, boundary)
    for _ in range(30):
        step_both(ctx, o, timer)
    ctx.clear_cached()
    o.clear_cached()
    for _ in range(10):
        step_both(ctx, o, timer)
    compare(ctx, o, "after clear")
    # grow: the host re-uploads its (sorted) arrays with new particles appended, like the shim does
    d = ctx.download()
    extra = (np.array([[0.3, 2.0]], np.float32) + np.stack(np.meshgrid(np.arange(12), np.arange(12)), -1).reshape(-1, 2).astype(np.float32)
             * np.float32(0.011))
    pos2 = np.concatenate([d["pos"], extra]).astype(np.float32)
    vel2 = np.concatenate([d["vel"], np.zeros_like(extra)]).astype(np.float32)
    ctx.upload(pos2, vel2)
    o.set_particles(pos2, vel2)
    for _ in range(40):
        step_both(ctx, o, timer)
    d, po = ctx.download(), o.positions()
    assert_bits_equal(d["pos"], po, "after growth positions")
    assert_bits_equal(d["vel"], o.velocities(), "after growth velocities")


def test_wcsph_host_mirror_solver():
    """Box<dyn Solver> = WCSPHSolver through the C++ mirror (world + timer + solver), vs the raw two-phase calls."""
    w = y.FluidParticleWorld()
    w.reset_fluid(1.0)
    solver = y.WCSPHSolver(w)
    tm = wcsph_timer()
    ctx = y.SphxContext()
    ctx.set_boundary(w.boundary_particles)
    ctx.upload(w.positions)
    t2 = wcsph_timer()
    for _ in range(50):
        solver.simulation_step(w, tm)
        vmax = ctx.wcsph_step_begin(t2.simulation_step())
        ctx.wcsph_step_finish(y.duration_as_secs_f32(t2.update_simulation_step(DIAM, vmax)))
    assert tm.simulation_step_ns() == t2.simulation_step_ns()
    d = ctx.download()
    assert_bits_equal(w.positions, d["pos"], "mirror positions")
    assert_bits_equal(w.velocities, d["vel"], "mirror velocities")


def test_wcsph_api_errors():
    ctx = y.SphxContext()
    with pytest.raises(y.SphxError):
        ctx.wcsph_step_begin(1e-3)  # nothing uploaded
    pos, boundary = dam_break(1.0)
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    with pytest.raises(y.SphxError):
        ctx.wcsph_step_finish(1e-3)  # no open step
    ctx.wcsph_step_begin(1e-3)
    with pytest.raises(y.SphxError):
        ctx.step_finish(1e-3)  # a WCSPH step is open, not a DFSPH one
    ctx.wcsph_step_finish(1e-3)
