"""Parity tests proper: the HIP path (through the C ABI, include/sphx.h) against the CPU oracle on the same inputs.

Bar: cell arrays, sorted order, neighbour counts and neighbour indices bit-exact; fp32 positions,
velocities, densities, alpha, kappa bit-identical too (the kernels keep the reference's operation order, un-fused, with
correctly rounded div/sqrt), so the tolerance written here is ZERO ulp.  The only values that may differ are the two
residual averages when an f64 partial-sum rounding lands on an f32 tie (never observed; reported, compared at 1 ulp).
"""
import numpy as np
import pytest
from util import assert_bits_equal, assert_same_neighbors, bench_world, brute_force_neighbors, dam_break, uniform_points

import yasph2d_amd as y
from oracle.oracle import KERNEL_POLY6, KERNEL_SPIKY, KERNEL_WENDLAND, Oracle

pytestmark = pytest.mark.gpu


def make_pair(pos, boundary=None, search_radius=None, fixed=(0, 0), list_span_limit=0):
    p = y.default_params(fixed_iterations=fixed)
    p.list_span_limit = list_span_limit
    if search_radius is not None:
        # NeighborhoodSearch::new(radius) of the criterion bench: radius = cell size = smoothing length
        p.smoothing_length = search_radius
    ctx = y.SphxContext(p)
    o = Oracle(search_radius=search_radius or 0.0)
    o.set_fixed_iterations(*fixed)
    if boundary is not None:
        ctx.set_boundary(boundary)
        o.set_boundary(boundary)
    ctx.upload(pos)
    o.set_particles(pos)
    return ctx, o


def compare_grid(ctx, o):
    d = ctx.download()
    assert_bits_equal(d["pos"], o.positions(), "sorted positions")
    np.testing.assert_array_equal(d["ids"], o.ids())
    for static in (False, True):
        f1, c1 = ctx.download_cells(static)
        f2, c2 = o.cells(static)
        np.testing.assert_array_equal(c1, c2)
        np.testing.assert_array_equal(f1, f2)
    bxy, bid = ctx.download_boundary()
    assert_bits_equal(bxy, o.boundary(), "sorted boundary")
    np.testing.assert_array_equal(bid, o.boundary_ids())
    assert_same_neighbors(ctx.download_neighbors(), o.neighbors())


def test_neighbor_search_reference_property():
    """neighborhood_search.rs:530-556 on the HIP path: list == ascending brute force, 1000 points, density 10, R = 1."""
    pos = uniform_points(1000, 10.0, 123456789)
    ctx, o = make_pair(pos, search_radius=1.0)
    ctx.update_neighborhood()
    o.update_neighborhood()
    compare_grid(ctx, o)
    p = ctx.download()["pos"]
    counts, start, lists = ctx.download_neighbors()
    for i in range(len(p)):
        np.testing.assert_array_equal(lists[int(start[i]):int(start[i + 1])], brute_force_neighbors(p, 1.0, i))


def test_neighbor_search_bench_20000():
    """benches/benchmarks/neighborhood_search.rs:10-29."""
    pos = uniform_points(20000, 10.0, 123456789)
    ctx, o = make_pair(pos, search_radius=1.0)
    for _ in range(2):  # cold + warm (already sorted input)
        ctx.update_neighborhood()
        o.update_neighborhood()
        compare_grid(ctx, o)


@pytest.mark.parametrize("kind", [KERNEL_WENDLAND, KERNEL_POLY6, KERNEL_SPIKY])
def test_update_densities_bench_world(kind):
    """benches/benchmarks/update_densities.rs:72-130: 8100 fluid + ~4420 boundary."""
    pos, boundary = bench_world()
    ctx, o = make_pair(pos, boundary)
    ctx.update_neighborhood()
    o.update_neighborhood()
    compare_grid(ctx, o)
    ctx.update_densities(kind)
    o.update_densities(kind)
    assert_bits_equal(ctx.download()["density"], o.densities(), f"densities kind={kind}")
    ctx.compute_alpha()
    o.compute_alpha()
    assert_bits_equal(ctx.download_solver_state()["alpha"], o.alpha(), "alpha")


def run_steps(ctx, o, steps, check_every=1, use_law=True, timer=None):
    """use_law: sphx_step_begin_law — the device derives dt from its own vmax and runs ahead of the host; step_finish fails
    unless the host timer arrives at the same bits, so every step also checks the device restatement of the timer law."""
    timer = timer if timer is not None else y.TimeManager()
    diam = np.float32(2.0) * np.float32(0.005)
    all_stats = []
    for s in range(steps):
        dt_prev = timer.simulation_step()
        vmax = ctx.step_begin(dt_prev, timer.law(diam) if use_law else None)
        dt_ns = timer.update_simulation_step(diam, vmax)
        st = ctx.step_finish(y.duration_as_secs_f32(dt_ns))
        so = o.dfsph_step()
        all_stats.append(st)
        assert o.timer_step_ns() == dt_ns, f"step {s}: dt differs"
        assert np.float32(vmax) == np.float32(so["vmax"]), f"step {s}: vmax {vmax} vs {so['vmax']}"
        for k in ("density_iterations", "divergence_iterations", "warmstart_density", "warmstart_divergence", "neighbor_entries"):
            assert st[k] == so[k], f"step {s}: {k} {st[k]} vs {so[k]}"
        for k in ("avg_density_error", "avg_divergence"):
            a, b = np.float32(st[k]), np.float32(so[k])
            assert a == b or abs(a - b) <= np.spacing(max(abs(a), abs(b))), f"step {s}: {k} {a} vs {b}"
        if (s + 1) % check_every == 0 or s == steps - 1:
            d = ctx.download()
            assert_bits_equal(d["pos"], o.positions(), f"step {s} positions")
            assert_bits_equal(d["vel"], o.velocities(), f"step {s} velocities")
            assert_bits_equal(d["density"], o.densities(), f"step {s} densities")
            np.testing.assert_array_equal(d["ids"], o.ids())
            ss = ctx.download_solver_state()
            assert_bits_equal(ss["alpha"], o.alpha(), f"step {s} alpha")
            assert_bits_equal(ss["kappa"], o.kappa(), f"step {s} kappa")
            assert_bits_equal(ss["stiffness"], o.stiffness(), f"step {s} stiffness")
    assert_same_neighbors(ctx.download_neighbors(), o.neighbors())
    return all_stats


def test_dfsph_dam_break_adaptive_400_steps():
    """The reference scene (4050 fluid + 6840 boundary, main.rs:177-196), adaptive timer, through free fall, impact and
    the first splash: every step's dt, vmax, iteration counts and every 20th step's full state must match the oracle."""
    pos, boundary = dam_break(1.0)
    ctx, o = make_pair(pos, boundary)
    run_steps(ctx, o, 400, check_every=20)


def test_dfsph_dam_break_fixed_iterations():
    """Fixed-iteration (parity) mode: 3 density + 2 divergence iterations per step exercises the warm-start passes."""
    pos, boundary = dam_break(1.0)
    ctx, o = make_pair(pos, boundary, fixed=(3, 2))
    run_steps(ctx, o, 60, check_every=10)


def test_dfsph_host_driven_dt():
    """Plain sphx_step_begin (the device waits for the host's dt): same results as the run-ahead path."""
    pos, boundary = dam_break(1.0)
    ctx, o = make_pair(pos, boundary)
    run_steps(ctx, o, 250, check_every=50, use_law=False)


def test_target_frame_timer_with_device_law():
    """TargetFrameLength timer (timemanager.rs:268-274): the lower bound the host passes in the law changes every step."""
    pos, boundary = dam_break(1.0)
    ctx, o = make_pair(pos, boundary)
    timer = y.TimeManager()
    target = 2_000_000  # 2 ms frames: the bound drops below timestep_min regularly
    timer.set_target_frame(target)
    o.timer_target_frame(target)
    diam = np.float32(0.01)
    for s in range(300):
        timer.on_step_started()
        o.timer_on_step_started()
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(diam))
        dt_ns = timer.update_simulation_step(diam, vmax)
        ctx.step_finish(y.duration_as_secs_f32(dt_ns))
        o.dfsph_step()
        assert o.timer_step_ns() == dt_ns, s
    d = ctx.download()
    assert_bits_equal(d["pos"], o.positions(), "positions")
    assert_bits_equal(d["vel"], o.velocities(), "velocities")


def test_timer_law_mismatch_is_detected():
    """A law that is not the one the host timer applies: step_finish refuses the host's dt and asks for a new upload."""
    pos, boundary = dam_break(1.0)
    ctx = y.SphxContext()
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    timer = y.TimeManager()
    diam = np.float32(0.01)
    for _ in range(150):  # into the CFL-limited regime, where the cfl factor matters
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(diam))
        ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(diam, vmax)))
    law = timer.law(diam)
    law.cfl_factor = 0.5
    vmax = ctx.step_begin(timer.simulation_step(), law)
    dt = y.duration_as_secs_f32(timer.update_simulation_step(diam, vmax))
    with pytest.raises(y.SphxError) as e:
        ctx.step_finish(dt)
    assert e.value.code == y._lib.ERR_INVALID_ARGUMENT
    with pytest.raises(y.SphxError):
        ctx.step_begin(timer.simulation_step())  # not ready until the state is uploaded again
    ctx.upload(pos)
    ctx.step_begin(timer.simulation_step())
    ctx.step_finish(dt)
    # a fixed-step timer goes through the same path
    ft = y.TimeManager(fixed_ns=500_000)
    vmax = ctx.step_begin(ft.simulation_step(), ft.law(diam))
    ctx.step_finish(y.duration_as_secs_f32(ft.update_simulation_step(diam, vmax)))


@pytest.mark.parametrize("span", [y.LISTS_32BIT, 60, 130])
def test_list_formats_do_not_change_results(span):
    """Workgroup-local neighbour lists: all workgroups on 32-bit global slots / a mix (a 256-particle workgroup has ~60-200 out-of-
    window neighbour entries at this size — more along the boundary — so table limits of 60 and 130 put most or some workgroups
    on the 32-bit fallback; the default limit of 512 puts none there).  Lists, states and scalars stay bit-identical to the oracle,
    through free fall and the impact (static neighbours, > 12 neighbours per particle: entries past the staged rows)."""
    pos, boundary = dam_break(float(np.sqrt(20000 / 4050)))
    ctx, o = make_pair(pos, boundary, list_span_limit=span)
    ctx.update_neighborhood()
    o.update_neighborhood()
    compare_grid(ctx, o)
    pos, boundary = dam_break(1.0)
    ctx, o = make_pair(pos, boundary, list_span_limit=span)
    run_steps(ctx, o, 330, check_every=55)


def test_two_pass_scan_switch(monkeypatch):
    """SPHX_SCAN_TWO_PASS=1 selects the two-launch cell-table scan (kept for A/B timing); the default is the one-launch
    look-back scan.  Both must give the oracle's grid and states — at 40 k particles the table spans several scan tiles, so the
    look-back really crosses workgroups."""
    s = float(np.sqrt(40000 / 4050))
    pos, boundary = dam_break(s)
    monkeypatch.setenv("SPHX_SCAN_TWO_PASS", "1")
    ctx, o = make_pair(pos, boundary)
    ctx.update_neighborhood()
    o.update_neighborhood()
    compare_grid(ctx, o)
    run_steps(ctx, o, 12, check_every=6)


def test_fused_count_switch(monkeypatch):
    """The last density correction of a step also does the advection + cell count of the re-grid that follows (CountArgs);
    SPHX_NO_FUSED_COUNT=1 keeps the separate count pass.  Both must follow the oracle bit for bit — with the adaptive loop, and
    with 3 fixed density iterations per step, where every correction counts and the first two counts of each step are thrown
    away (the histogram is cleared again)."""
    pos, boundary = dam_break(1.0)
    for fixed, steps in (((0, 0), 120), ((3, 2), 60)):  # (the fixed-iteration scene blows up after ~100 steps)
        monkeypatch.setenv("SPHX_NO_FUSED_COUNT", "1")
        ctx, o = make_pair(pos, boundary, fixed=fixed)
        stats = run_steps(ctx, o, steps, check_every=20, use_law=fixed == (0, 0))
        monkeypatch.delenv("SPHX_NO_FUSED_COUNT")
        ctx2, o2 = make_pair(pos, boundary, fixed=fixed)
        stats2 = run_steps(ctx2, o2, steps, check_every=20, use_law=fixed == (0, 0))
        assert [s["density_iterations"] for s in stats] == [s["density_iterations"] for s in stats2]
        if fixed[0]:
            assert stats2[-1]["density_iterations"] == fixed[0]
        a, b = ctx.download(), ctx2.download()
        for k in ("pos", "vel", "density"):
            assert_bits_equal(a[k], b[k], k)


def test_the_walks_forms_for_stale_lists_follow_the_oracle_too(monkeypatch):
    """The walks have two arms: FAST — exact rsq-based square root, no min(q, 1) — while the lists belong to the
    positions, and plain sqrtf + clamp after positions were replaced behind the lists' back (or SPHX_QCLAMP=1).  Both must follow
    the oracle bit for bit: the switch through the impact with warm starts, and the stale case itself — an upload of DIFFERENT
    positions with an unchanged particle count, after which the reference (and the oracle) walk the old lists over the new positions,
    coincident pairs and pairs beyond h included."""
    pos, boundary = dam_break(1.0)
    monkeypatch.setenv("SPHX_QCLAMP", "1")
    ctx, o = make_pair(pos, boundary)
    run_steps(ctx, o, 200, check_every=40)
    monkeypatch.delenv("SPHX_QCLAMP")
    ctx2, o2 = make_pair(pos, boundary)
    timer2 = y.TimeManager()
    run_steps(ctx2, o2, 200, check_every=40, timer=timer2)
    a, b = ctx.download(), ctx2.download()
    for k in ("pos", "vel", "density"):
        assert_bits_equal(a[k], b[k], k)
    # stale lists: shuffle the positions (every list now points at particles somewhere else), put two particles on one spot
    rng = np.random.default_rng(3)
    p2 = np.ascontiguousarray(b["pos"][rng.permutation(len(pos))])
    p2[1] = p2[0]
    ctx2.upload(p2, b["vel"])
    o2.set_particles(p2, b["vel"])
    # (densities are outputs — the oracle's set_particles zeroes them, the device keeps the old ones: recompute them on both sides,
    # through the stale lists, before the step reads them in the XSPH term)
    ctx2.update_densities(KERNEL_WENDLAND)
    o2.update_densities(KERNEL_WENDLAND)
    assert_bits_equal(ctx2.download()["density"], o2.densities(), "densities through stale lists")
    run_steps(ctx2, o2, 3, check_every=1, timer=timer2)


def test_arrival_slots_that_do_not_fit_the_cell_word_go_through_the_side_array(monkeypatch):
    """The re-grid keeps a particle's cell index and its arrival slot inside the cell in one 32-bit word (count_cell, GridView::cbits);
    a slot that does not fit the bits the table size leaves goes to a side array.  With SPHX_CBITS_MIN=30 the word has room for the
    slots 0..2: every cell with four particles (most cells of the scene) takes the side path — same run as the oracle, bit for bit."""
    pos, boundary = dam_break(1.0)
    monkeypatch.setenv("SPHX_CBITS_MIN", "30")
    ctx, o = make_pair(pos, boundary)
    run_steps(ctx, o, 120, check_every=40)
    monkeypatch.delenv("SPHX_CBITS_MIN")


def test_dfsph_scale_40k():
    s = float(np.sqrt(40000 / 4050))
    pos, boundary = dam_break(s)
    ctx, o = make_pair(pos, boundary)
    run_steps(ctx, o, 30, check_every=10)


def test_clear_cached_and_reupload():
    """Solver::clear_cached_data (dfsph.rs:406-412) then a fresh scene: the warm-up block runs again."""
    pos, boundary = dam_break(1.0)
    ctx, o = make_pair(pos, boundary)
    run_steps(ctx, o, 5)
    ctx.clear_cached()
    o.clear_cached()
    ctx.upload(pos)
    o.set_particles(pos)
    t = y.TimeManager()
    o.timer_adaptive(t.timestep_max_ns, t.timestep_min_ns, 1.5)  # time_manager.restart() (main.rs:296)
    run_steps(ctx, o, 5)


def test_empty_and_tiny():
    ctx = y.SphxContext()
    ctx.upload(np.zeros((0, 2), np.float32))
    assert ctx.step_begin(1e-4) == 0.0
    ctx.step_finish(1e-4)
    ctx2, o = make_pair(np.array([[0.5, 0.5], [0.505, 0.5], [0.5, 0.507]], np.float32))
    run_steps(ctx2, o, 3)


def test_neighbor_cap_flag():
    rng = np.random.default_rng(11)
    pos = (np.float32(5.0) + rng.random((100, 2), dtype=np.float32) * np.float32(0.2)).astype(np.float32)
    ctx, o = make_pair(pos, search_radius=1.0)
    ctx.update_neighborhood()
    o.update_neighborhood()
    assert_same_neighbors(ctx.download_neighbors(), o.neighbors())
    assert (ctx.download_neighbors()[0][:, 1] == 64).all()


def test_error_codes():
    from yasph2d_amd import _lib

    ctx = y.SphxContext()
    with pytest.raises(y.SphxError) as e:
        ctx.step_begin(1e-4)
    assert e.value.code == _lib.ERR_NOT_READY
    ctx.upload(np.array([[0.0, 0.0]], np.float32))
    with pytest.raises(y.SphxError) as e:
        ctx.step_finish(1e-4)
    assert e.value.code == _lib.ERR_NOT_READY
    with pytest.raises(y.SphxError) as e:
        ctx.step_begin(float("nan"))
    assert e.value.code == _lib.ERR_INVALID_ARGUMENT


def test_host_mirror_solver_interface():
    """Solver::simulation_step(&mut world, &mut time_manager) through the C++ host mirror == the oracle stepping."""
    w = y.FluidParticleWorld()
    w.reset_fluid(1.0)
    o = Oracle()
    o.set_boundary(w.boundary_particles)
    o.set_particles(w.positions)
    solver = y.DFSPHSolver(w)
    t = y.TimeManager()
    for s in range(12):
        st = solver.simulation_step(w, t, sync_world=True)
        so = o.dfsph_step()
        assert t.simulation_step_ns() == o.timer_step_ns()
        assert st["density_iterations"] == so["density_iterations"]
    assert_bits_equal(w.positions, o.positions(), "world positions")
    assert_bits_equal(w.velocities, o.velocities(), "world velocities")
    assert_bits_equal(w.densities, o.densities(), "world densities")
    np.testing.assert_array_equal(w.particle_ids, o.ids())
    assert t.num_steps == 12
