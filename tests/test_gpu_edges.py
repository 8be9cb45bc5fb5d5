"""Edge cases of the path on the GPU, each against the oracle: particle-count changes mid-run (the reference's resize keeps
the slot-bound warm-start values, dfsph.rs:419-428), boundary replaced mid-run, static neighbours next to the 64 cap, fixed
time step, error codes for non-finite input, and a long violent run with warm starts and many solver iterations."""
import numpy as np
import pytest
from util import assert_bits_equal, assert_same_neighbors, dam_break

import yasph2d_amd as y
from oracle.oracle import Oracle
from yasph2d_amd import _lib

pytestmark = pytest.mark.gpu


def pair(pos, boundary=None, vel=None):
    ctx, o = y.SphxContext(), Oracle()
    if boundary is not None:
        ctx.set_boundary(boundary)
        o.set_boundary(boundary)
    ctx.upload(pos, vel)
    o.set_particles(pos, vel)
    return ctx, o


def step_both(ctx, o, timer, check=True):
    vmax = ctx.step_begin(timer.simulation_step(), timer.law(np.float32(0.01)))
    dt_ns = timer.update_simulation_step(np.float32(0.01), vmax)
    st = ctx.step_finish(y.duration_as_secs_f32(dt_ns))
    so = o.dfsph_step()
    if check:
        assert dt_ns == o.timer_step_ns()
        for k in ("density_iterations", "divergence_iterations", "warmstart_density", "warmstart_divergence"):
            assert st[k] == so[k], (k, st[k], so[k])
    return st, so


def compare_state(ctx, o, what=""):
    d = ctx.download()
    assert_bits_equal(d["pos"], o.positions(), what + " positions")
    assert_bits_equal(d["vel"], o.velocities(), what + " velocities")
    assert_bits_equal(d["density"], o.densities(), what + " densities")
    ss = ctx.download_solver_state()
    assert_bits_equal(ss["kappa"], o.kappa(), what + " kappa")
    assert_bits_equal(ss["stiffness"], o.stiffness(), what + " stiffness")
    assert_same_neighbors(ctx.download_neighbors(), o.neighbors())


def test_violent_phase_with_warm_starts():
    """900 adaptive steps of the reference scene: impact, splash-up, many steps with Iv > 1 and warm starts."""
    pos, boundary = dam_break(1.0)
    ctx, o = pair(pos, boundary)
    timer = y.TimeManager()
    warm = multi = 0
    for s in range(900):
        st, _ = step_both(ctx, o, timer)
        warm += st["warmstart_density"] + st["warmstart_divergence"]
        multi += (st["density_iterations"] > 1) + (st["divergence_iterations"] > 1)
        if s % 150 == 149:
            compare_state(ctx, o, f"step {s}")
    assert warm > 20 and multi > 20, (warm, multi)
    compare_state(ctx, o, "final")


def test_adding_particles_mid_run_keeps_slot_bound_warm_start_values():
    """add_fluid_rect during the run (count changes -> warm-up block again, Vec::resize keeps old entries, dfsph.rs:420-422)."""
    w = y.FluidParticleWorld()
    w.reset_fluid(1.0)
    pos, boundary = w.positions, w.boundary_particles
    ctx, o = pair(pos, boundary)
    timer = y.TimeManager()
    for _ in range(150):  # through the impact so that kappa / stiffness are non-zero
        step_both(ctx, o, timer)
    assert np.abs(ctx.download_solver_state()["stiffness"]).max() > 0
    # the application appends a block of fluid to the (sorted) arrays it got back
    d = ctx.download()
    extra = (np.array([1.2, 1.0], np.float32) + np.stack(np.meshgrid(np.arange(20), np.arange(20)), -1).reshape(-1, 2).astype(np.float32) * np.float32(0.0111)).astype(np.float32)
    new_pos = np.concatenate([d["pos"], extra])
    new_vel = np.concatenate([d["vel"], np.zeros_like(extra)])
    ctx.upload(new_pos, new_vel)
    o.set_particles(new_pos, new_vel)
    for _ in range(40):
        st, _ = step_both(ctx, o, timer)
    compare_state(ctx, o, "after growth")


def test_boundary_replaced_mid_run():
    pos, boundary = dam_break(1.0)
    ctx, o = pair(pos, boundary)
    timer = y.TimeManager()
    for _ in range(20):
        step_both(ctx, o, timer)
    b2 = boundary[boundary[:, 1] < 2.0]  # drop the lid: boundary_changed = true (fluidparticleworld.rs:194)
    ctx.set_boundary(b2)
    o.set_boundary(b2)
    for _ in range(20):
        step_both(ctx, o, timer)
    compare_state(ctx, o, "after boundary change")
    bxy, _ = ctx.download_boundary()
    assert_bits_equal(bxy, o.boundary(), "sorted boundary")


def test_boundary_grown_mid_run_keeps_the_old_tail_until_the_regrid():
    """sphx_set_boundary with MORE particles than the arrays hold reallocates the [N|B] arrays between two steps.  The half step in
    front of the next re-grid still walks the previous lists (dfsph.rs:436-497 run before update_neighborhood_datastructure, :512):
    their static entries must find the previous boundary's records where they were (round-5 advisor finding: the tail was left
    uninitialised and the FAST walk flag stayed set).  The scene is built so that the reference does the same thing: the boundary is
    handed over in its sorted order (sorting it again is the identity) and the new particles are appended BEHIND it, so the old
    static indices name the same particles in the new array."""
    pos, boundary = dam_break(1.0)
    ctx0 = y.SphxContext()
    ctx0.set_boundary(boundary)
    ctx0.upload(pos)
    ctx0.update_neighborhood()
    sorted_b, _ = ctx0.download_boundary()
    ctx0.close()
    ctx, o = pair(pos, sorted_b)
    timer = y.TimeManager()
    for _ in range(30):
        step_both(ctx, o, timer)
    bxy, _ = ctx.download_boundary()
    assert_bits_equal(bxy, sorted_b, "a sorted boundary stays as it is")
    xs = np.arange(0.3, 1.7, 0.01, dtype=np.float32)
    shelf = np.stack([xs, np.full_like(xs, np.float32(0.35))], -1)  # a shelf across the tank, in the fluid's way
    b2 = np.concatenate([sorted_b, shelf, shelf + np.float32([0.0, 0.01])]).astype(np.float32)
    assert len(b2) > len(sorted_b)
    ctx.set_boundary(b2)
    o.set_boundary(b2)
    for s in range(40):
        step_both(ctx, o, timer)
        if s in (0, 1, 39):
            compare_state(ctx, o, f"step {s} after the boundary grew")
    bxy, _ = ctx.download_boundary()
    assert_bits_equal(bxy, o.boundary(), "sorted boundary")


def test_fixed_time_step():
    pos, boundary = dam_break(1.0)
    ctx, o = pair(pos, boundary)
    timer = y.TimeManager(fixed_ns=500_000)  # SimulationStepConfig::FixedTimeStep (timemanager.rs:40)
    o.timer_fixed(500_000)
    for _ in range(60):
        step_both(ctx, o, timer)
    compare_state(ctx, o, "fixed dt")


def test_dense_blob_next_to_a_wall_hits_the_neighbor_cap():
    """~90 particles within one radius next to boundary particles: dynamic lists are capped at 64 and the reference would panic on
    the first static hit (neighborhood_search.rs:373) -> SPHX_ERR_NEIGHBOR_PANIC, not a crash."""
    rng = np.random.default_rng(4)
    pos = (np.array([0.5, 0.5], np.float32) + rng.random((90, 2), dtype=np.float32) * np.float32(0.012)).astype(np.float32)
    wall = np.stack([np.linspace(0.49, 0.53, 9, dtype=np.float32), np.full(9, 0.495, np.float32)], -1)
    ctx = y.SphxContext()
    ctx.set_boundary(wall)
    ctx.upload(pos)
    with pytest.raises(y.SphxError) as e:
        ctx.update_neighborhood()
    assert e.value.code == _lib.ERR_NEIGHBOR_PANIC
    # without the wall: capped lists, flag only
    ctx2, o2 = pair(pos)
    ctx2.update_neighborhood()
    o2.update_neighborhood()
    assert_same_neighbors(ctx2.download_neighbors(), o2.neighbors())
    assert o2.neighbor_flags() & 1


def test_nonfinite_velocity_is_an_error_code():
    pos, boundary = dam_break(1.0)
    vel = np.zeros_like(pos)
    vel[17] = [np.inf, 0.0]
    ctx = y.SphxContext()
    ctx.set_boundary(boundary)
    ctx.upload(pos, vel)
    with pytest.raises(y.SphxError) as e:
        ctx.step_begin(1e-4)
    assert e.value.code == _lib.ERR_NONFINITE  # Duration::from_secs_f32 would panic (timemanager.rs:264)


def test_two_contexts_are_independent():
    pos, boundary = dam_break(1.0)
    a, oa = pair(pos, boundary)
    b, ob = pair(pos[::2].copy(), boundary)
    ta, tb = y.TimeManager(), y.TimeManager()
    for _ in range(10):
        step_both(a, oa, ta)
        step_both(b, ob, tb)
    compare_state(a, oa, "ctx a")
    compare_state(b, ob, "ctx b")


def test_viewer_feed_strided_async_download():
    """sphx_view_request / sphx_view_fetch: {x, y, |v|} of every stride-th particle, copied on a separate
    stream while the simulation keeps stepping; equals the full download taken at the time of the request."""
    pos, boundary = dam_break(2.0)
    ctx = y.SphxContext()
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    timer = y.TimeManager()
    diam = np.float32(0.01)

    def step():
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(diam))
        ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(diam, vmax)))

    with pytest.raises(y.SphxError):
        ctx.view_fetch()  # nothing requested yet
    for _ in range(20):
        step()
    for stride in (1, 7, 1000, 10**9):
        d = ctx.download()
        m = ctx.view_request(stride)
        assert m == (len(pos) + stride - 1) // stride
        for _ in range(3):  # the copy overlaps these steps; the snapshot is the state at request time
            step()
        v = ctx.view_fetch()
        assert v.shape == (m, 3)
        assert_bits_equal(v[:, :2], d["pos"][::stride], f"stride {stride} positions")
        speed = np.sqrt(d["vel"][::stride, 0] * d["vel"][::stride, 0] + d["vel"][::stride, 1] * d["vel"][::stride, 1])
        assert_bits_equal(v[:, 2], speed.astype(np.float32), f"stride {stride} speed")
    # polling form: eventually ready
    ctx.view_request(3)
    got = None
    for _ in range(100000):
        got = ctx.view_fetch(wait=False)
        if got is not None:
            break
    assert got is not None and got.shape[1] == 3


def test_particles_that_outrun_the_cell_directory_are_kept():
    """A blob moving 150 cells per step (3000 m/s at a fixed 1 ms step) leaves the covered 64x64-cell blocks in one step.  The
    reference has no such table and simply carries on; so does the device now: nothing is lost, SPHX_FLAG_STRAY_PARTICLES is
    reported, the directory is re-covered, and the blob arrives where ballistic motion takes it."""
    side = 30
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2).astype(np.float32)
    pos = (np.float32(1.0) + g * np.float32(0.0111)).astype(np.float32)
    slow = (np.float32(-3.0) + g * np.float32(0.0111)).astype(np.float32)  # a second blob that stays put
    allpos = np.concatenate([pos, slow])
    vel = np.concatenate([np.tile(np.array([[3000.0, 0.0]], np.float32), (len(pos), 1)), np.zeros_like(slow)])
    p = y.default_params()
    p.gravity[0], p.gravity[1] = 0.0, 0.0
    ctx = y.SphxContext(p)
    ctx.upload(allpos, vel)
    timer = y.TimeManager(fixed_ns=1_000_000)
    diam = np.float32(0.01)
    flagged = 0
    for _ in range(30):
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(diam))
        st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(diam, vmax)))
        flagged += bool(st["flags"] & y.FLAG_STRAY_PARTICLES)
    assert flagged >= 1
    d = ctx.download()
    assert sorted(d["ids"].tolist()) == list(range(len(allpos)))
    assert np.isfinite(d["pos"]).all()
    fast = d["ids"] < len(pos)
    travelled = d["pos"][fast, 0].mean() - pos[:, 0].mean()
    assert abs(travelled - 3000.0 * 30e-3) < 0.05 * 90.0  # ~90 m, ballistic
    assert np.ptp(d["pos"][fast, 0]) < 2.0 and np.ptp(d["pos"][fast, 1]) < 2.0  # still a blob
    assert abs(d["pos"][~fast, 0].mean() - slow[:, 0].mean()) < 0.1  # the other blob stayed where it was
    # both blobs have their neighbours again
    counts, _, _ = ctx.download_neighbors()
    assert counts[:, 1].mean() > 5


def test_far_apart_pools_use_a_sparse_directory():
    """Two pools 1.2 km apart (60 000 cells): a dense rectangle of 64x64-cell blocks over their bounding box would need > 3 G table
    entries; the sparse directory covers the two pools and their fringes only.  Results equal the oracle's."""
    side = 40
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2).astype(np.float32)
    a = (np.array([-90.0, -90.0], np.float32) + g * np.float32(0.0111)).astype(np.float32)
    b = (np.array([1100.0, 1100.0], np.float32) + g * np.float32(0.0111)).astype(np.float32)
    pos = np.concatenate([a, b])
    ctx, o = pair(pos, None)
    timer = y.TimeManager()
    for _ in range(30):
        step_both(ctx, o, timer)
    compare_state(ctx, o, "two pools")


def test_device_run_solver_loops_equal_host_run_loops(monkeypatch):
    """The residual test of dfsph.rs:221-236 / :376-391 runs on the device (LoopArgs: iterations queued ahead, the terminating one
    marks the loop done, the rest return at once); SPHX_HOST_LOOP=1 keeps the host in the loop.  Same bits, same counts — through
    the impact of the reference scene, where the counts jump up and down (under- and over-predicted queues) and warm starts fire;
    and with tolerances tight enough that the density loop iterates too."""
    pos, boundary = dam_break(1.0)

    def run(host_loop, tight):
        if host_loop:
            monkeypatch.setenv("SPHX_HOST_LOOP", "1")
        else:
            monkeypatch.delenv("SPHX_HOST_LOOP", raising=False)
        p = y.default_params()
        if tight:
            p.max_avg_density_error = np.float32(1e-8)
            p.max_divergence_error = np.float32(1e-6)
            p.max_density_iterations = 7   # the cap (+1) is reached: dfsph.rs:236
        ctx = y.SphxContext(p)
        ctx.set_boundary(boundary)
        ctx.upload(pos)
        timer = y.TimeManager()
        counts = []
        for _ in range(120 if tight else 450):
            vmax = ctx.step_begin(timer.simulation_step(), timer.law(np.float32(0.01)))
            st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(np.float32(0.01), vmax)))
            counts.append((st["density_iterations"], st["divergence_iterations"], st["warmstart_density"], st["warmstart_divergence"], st["flags"] & 6,
                           np.float32(st["avg_density_error"]).tobytes(), np.float32(st["avg_divergence"]).tobytes()))
        return ctx.download(), ctx.download_solver_state(), counts

    for tight in (False, True):
        da, sa, ca = run(False, tight)
        db, sb, cb = run(True, tight)
        assert ca == cb
        assert max(c[1] for c in ca) > 3 and sum(c[2] + c[3] for c in ca) > 10
        if tight:
            assert max(c[0] for c in ca) == 8 and any(c[4] & 2 for c in ca), "the density loop must have hit its iteration cap"
        for k in ("pos", "vel", "density"):
            assert_bits_equal(da[k], db[k], k)
        for k in ("kappa", "stiffness", "alpha"):
            assert_bits_equal(sa[k], sb[k], k)


def test_run_ahead_over_the_step_boundary_changes_nothing(monkeypatch):
    """sphx_step_finish queues the next step's non-pressure pass behind the divergence iterations it predicts (sphx_ctx::ahead); the
    next sphx_step_begin adopts it only if nothing touched the context and dt_prev is the dt it was queued with.  Same bits as with
    SPHX_RUN_AHEAD=0 through the impact (under-predicted loops: the queued pass is discarded), across a boundary edit, a re-upload,
    a clear_cached_data and a caller that passes a different dt_prev."""
    pos, boundary = dam_break(1.0)

    def run(ahead):
        monkeypatch.setenv("SPHX_RUN_AHEAD", "1" if ahead else "0")
        ctx = y.SphxContext(y.default_params())
        ctx.set_boundary(boundary)
        ctx.upload(pos)
        timer = y.TimeManager()
        d = np.float32(0.01)
        counts = []

        def steps(k, dt_prev_override=None):
            for j in range(k):
                dtp = timer.simulation_step() if dt_prev_override is None or j else dt_prev_override
                vmax = ctx.step_begin(dtp, timer.law(d))
                st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(d, vmax)))
                counts.append((st["density_iterations"], st["divergence_iterations"], np.float32(st["vmax"]).tobytes()))

        steps(300)                                   # through the impact
        ctx.set_boundary(boundary[: len(boundary) // 2 * 2 - 40])  # a gap in the wall
        steps(20)
        steps(5, dt_prev_override=np.float32(0.0007))  # not the dt the previous finish ran with
        ctx.clear_cached()
        steps(10)
        st = ctx.download()
        keep = st["ids"] % 3 != 0
        o = np.argsort(st["ids"][keep])
        ctx.upload(st["pos"][keep][o], st["vel"][keep][o])
        steps(20)
        return ctx.download(), counts

    a, ca = run(True)
    b, cb = run(False)
    assert ca == cb and max(c[1] for c in ca) > 3
    assert np.array_equal(a["ids"], b["ids"])
    for k in ("pos", "vel", "density"):
        assert_bits_equal(a[k], b[k], k)


def test_cell_table_follows_the_fluid():
    """A blob in free fall crosses a 64x64-cell block every ~70 steps.  The block directory grows a ring whenever a particle reaches
    its fringe; every 256 builds the coverage is re-derived from the blocks that hold particles (cover_from_occupancy), so the table
    — what every build's histogram scan runs over — stays a few blocks instead of everything the blob ever came near.  Same bits as
    the oracle across the re-covers."""
    side = 24
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2).astype(np.float32)
    pos = (np.array([3.0, 40.0], np.float32) + g * np.float32(0.0111)).astype(np.float32)
    ctx, o = pair(pos, None)
    timer = y.TimeManager()
    first = None
    seen = []
    for s in range(1100):
        step_both(ctx, o, timer, check=(s % 50 == 0))
        if first is None:
            first = ctx.grid_info()["blocks"]
        seen.append(ctx.grid_info()["blocks"])
    compare_state(ctx, o, "falling blob")
    d = ctx.download()
    assert d["pos"][:, 1].max() < 40.0 - 4 * 1.28, "the blob must have fallen through several blocks"
    assert max(seen) > 2 * first, "rings were added on the way"
    assert min(seen[768:]) <= 16 and seen[-1] <= 36, (first, max(seen), seen[-1])


def test_a_cell_with_thousands_of_particles_does_not_stall_the_regrid():
    """Round-1 review: the stable rank of k_rank_gather is quadratic in the cell occupancy.  40 000 particles collapsed into ONE cell
    (1.6e9 dependent loads on the old path) must re-grid in milliseconds: past 4096 particles per cell the order inside the cell is
    the arrival order, the step reports SPHX_FLAG_DENSE_CELL, and nothing else changes — every particle is still there, cells and
    neighbour counts are what the oracle's are (the lists of such a cell hold the cap's worth of its mates)."""
    import time

    rng = np.random.default_rng(5)
    n = 40000
    blob = (np.array([2.001, 3.001], np.float32) + rng.random((n, 2), dtype=np.float32) * np.float32(0.017)).astype(np.float32)  # inside one 0.02 cell
    side = 30
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2).astype(np.float32)
    calm = (np.array([4.0, 3.0], np.float32) + g * np.float32(0.0111)).astype(np.float32)
    pos = np.concatenate([blob, calm])
    ctx = y.SphxContext()
    ctx.upload(pos)
    ctx.update_neighborhood()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.update_neighborhood()
    ctx.synchronize()
    assert time.perf_counter() - t0 < 0.5
    assert ctx.last_flags() & y.FLAG_DENSE_CELL
    d = ctx.download()
    assert np.array_equal(np.sort(d["ids"]), np.arange(len(pos)))
    assert_bits_equal(d["pos"][np.argsort(d["ids"])], pos, "positions follow their ids")
    first, cidx = ctx.download_cells()
    occ = np.diff(first)
    assert occ.max() == n and (occ[occ < n] <= 9).all()
    nb = ctx.download_neighbors()
    counts = nb["counts"] if isinstance(nb, dict) else nb[0]
    assert counts.reshape(-1, 2)[:, 1].max() == 64
    # the calm patch is ordinary fluid: bit-exact against the oracle on its own
    o = Oracle()
    o.set_particles(calm, None)
    o.update_neighborhood()
    ctx2 = y.SphxContext()
    ctx2.upload(calm)
    ctx2.update_neighborhood()
    assert not (ctx2.last_flags() & y.FLAG_DENSE_CELL)
    assert_same_neighbors(ctx2.download_neighbors(), o.neighbors())


def test_divergence_error_folded_into_the_neighbour_build_changes_nothing(monkeypatch):
    """When the divergence loop starts without a warm start, the neighbour build also does that loop's first compute_density_change
    (k_neighbor_build<2>: errors, err * alpha, zeroed warm-start stiffness, residual).  SPHX_FUSE_DIV=0 runs the separate kernel.
    Same bits and counts through the impact, where steps with and without a warm start alternate."""
    pos, boundary = dam_break(1.0)

    def run(fused):
        monkeypatch.setenv("SPHX_FUSE_DIV", "1" if fused else "0")
        ctx = y.SphxContext(y.default_params())
        ctx.set_boundary(boundary)
        ctx.upload(pos)
        timer = y.TimeManager()
        d = np.float32(0.01)
        counts = []
        for _ in range(420):
            vmax = ctx.step_begin(timer.simulation_step(), timer.law(d))
            st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(d, vmax)))
            counts.append((st["density_iterations"], st["divergence_iterations"], st["warmstart_divergence"], np.float32(st["avg_divergence"]).tobytes()))
        return ctx.download(), ctx.download_solver_state(), counts

    a, sa, ca = run(True)
    b, sb, cb = run(False)
    assert ca == cb
    assert any(c[2] for c in ca) and any(not c[2] for c in ca[100:]), "both kinds of step must occur"
    for k in ("pos", "vel", "density"):
        assert_bits_equal(a[k], b[k], k)
    for k in ("kappa", "stiffness", "alpha"):
        assert_bits_equal(sa[k], sb[k], k)


@pytest.mark.gpu
@pytest.mark.parametrize("tolerance_scale", [1.0, 0.1])
def test_prediction_folded_into_the_first_density_iteration_changes_nothing(monkeypatch, tolerance_scale):
    """With the timer law on the device and no density warm start ahead, the first compute_density_error of a step also does the
    velocity prediction (k_compute_error<false, true>: dt from the max-velocity reduction, neighbours staged as v + a dt, own result
    into the other velocity buffer).  SPHX_FUSE_PREDICT=0 runs the separate k_predict.  Same bits, time steps and counts — through
    free fall and the impact; a particle block added mid-run (warm-up block, changed count) included.  With the reference's
    tolerance the density loop of this scene never needs a second iteration (no density warm start ever: every step is a fused one);
    with a tenth of it steps with a density warm start occur (around the impact), and the separate k_predict takes over for those."""
    def run(fused):
        monkeypatch.setenv("SPHX_FUSE_PREDICT", "1" if fused else "0")
        w = y.FluidParticleWorld()
        w.reset_fluid(1.0)
        t = y.TimeManager()
        p = y.default_params(fixed_iterations=(0, 0))
        p.max_avg_density_error *= tolerance_scale
        s = y.DFSPHSolver(w, p)
        st = [s.simulation_step(w, t, sync_world=False) for _ in range(300)]
        s.sync_world(w)
        w.add_fluid_rect(1.2, 1.0, 0.2, 0.2, 0.05)
        st += [s.simulation_step(w, t, sync_world=False) for _ in range(60)]
        s.sync_world(w)
        return np.array(w.positions), np.array(w.velocities), np.array(w.densities), t.total_simulated_ns, [
            (x["density_iterations"], x["divergence_iterations"], x["warmstart_density"], np.float32(x["avg_density_error"]).tobytes()) for x in st]

    pa, va, da, ta, ca = run(True)
    pb, vb, db, tb, cb = run(False)
    assert ta == tb and ca == cb
    assert sum(1 for c in ca if not c[2]) > 100, "steps without a density warm start (the fused ones) must occur"
    if tolerance_scale < 1.0:
        assert any(c[2] for c in ca), "steps with a density warm start must occur"
    assert_bits_equal(pa, pb, "pos")
    assert_bits_equal(va, vb, "vel")
    assert_bits_equal(da, db, "density")
