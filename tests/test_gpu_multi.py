"""sphx_multi — the tile step loop inside libsphx (csrc/sphx_tiles.cpp) — against the reference implementation of the same loop:
tests/tiles_reference.py driving the CPU oracle (tests/tile_oracle_backend.py).  Same cuts, same halo rules, same re-partitioning:
every owned particle must agree BIT FOR BIT, as must iteration counts, dt, the number of halo exchanges and the final cuts.

  * all tiles in one process (sphx_multi_create: one host thread + HIP stream per tile, peer copies ordered by events);
  * one process per tile (sphx_multi_create_rank) with a caller-supplied communicator over torch.distributed/gloo;
  * the built-in RCCL transport is brought up with one rank (RCCL refuses two ranks on the one GPU of this box — gpurun_out/
    r02_probe1/rccl_same_gpu.log: "Duplicate GPU detected" — so its multi-rank send/recv cannot be exercised here).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
from test_tiles_cpu import GridLayout, run_tiles_threaded
from util import assert_bits_equal, dam_break

import yasph2d_amd as y
from yasph2d_amd import _lib
from yasph2d_amd.multi import MultiSolver
from tiles_reference import cell_coord, quantile_cuts

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def oracle_tiles(pos, boundary, world, axis, steps, **kw):
    from tile_oracle_backend import OracleTileBackend

    out, cuts = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, world, axis, steps, **kw)
    return out, list(run_tiles_threaded.final_cuts)


def by_id(d):
    o = np.argsort(d["ids"])
    return {k: v[o] for k, v in d.items()}


def merged_oracle(out):
    parts = [o[0] for o in out]
    d = {k: np.concatenate([p[k] for p in parts]) for k in ("ids", "pos", "vel", "density")}
    return by_id(d)


def compare(m_out, o_out, stats_m, stats_o):
    a, b = by_id(m_out), merged_oracle(o_out)
    np.testing.assert_array_equal(a["ids"], b["ids"])
    for k in ("pos", "vel", "density"):
        assert_bits_equal(a[k], b[k], k)
    for sm, so in zip(stats_m, stats_o):
        assert (sm["density_iterations"], sm["divergence_iterations"], sm["dt_ns"]) == (so["density_iterations"], so["divergence_iterations"], so["dt_ns"])


@pytest.mark.parametrize("world,axis,halo,fixed,steps", [(2, 1, 16, (0, 0), 120), (3, 1, 8, (0, 0), 60), (2, 0, 6, (0, 0), 60), (2, 1, 10, (3, 2), 80)])
def test_in_process_strips_bit_exact_vs_reference_loop(world, axis, halo, fixed, steps):
    pos, boundary = dam_break(1.0)
    cuts = quantile_cuts(cell_coord(pos, axis), world)
    o, _ = oracle_tiles(pos, boundary, world, axis, steps, halo=halo, fixed=fixed, cuts=cuts, adaptive_halo=True, rebalance_every=8)
    m = MultiSolver(y.default_params(fixed_iterations=fixed), devices=[0] * world, halo=halo, rebalance_every=8)
    m.set_strips(axis, cuts)
    m.set_boundary(boundary)
    m.upload(pos)
    timer = y.TimeManager()
    stats = [m.step(timer) for _ in range(steps)]
    compare(m.download(), o, stats, o[0][1])
    assert m.info()["exchanges"] == o[0][2]
    assert m.info()["transport"].startswith("in-process")


@pytest.mark.parametrize("overlap", [False, True])
def test_in_process_2x2_through_the_impact_bit_exact_vs_reference_loop(overlap):
    """SURVEY.md 8(e) "4 GPUs: 2x2 tiles": warm starts, diagonal migration, budget-triggered extra exchanges, re-partitioning.
    overlap: the halo records travel on a second stream while the tile counts the cells of the particles it kept."""
    pos, boundary = dam_break(2.0)
    lay = GridLayout.quantile(pos, 2, 2)
    kw = dict(halo=10, fixed=(3, 2), rebalance_every=4, layout=lambda: GridLayout(lay.xcuts, lay.ycuts), adaptive_halo=True)
    o, final = oracle_tiles(pos, boundary, 4, None, 150, **kw)
    m = MultiSolver(y.default_params(fixed_iterations=(3, 2)), devices=[0, 0, 0, 0], halo=10, rebalance_every=4, overlap_exchange=overlap)
    m.set_grid(lay.xcuts, lay.ycuts)
    m.set_boundary(boundary)
    m.upload(pos)
    timer = y.TimeManager()
    stats = [m.step(timer) for _ in range(150)]
    compare(m.download(), o, stats, o[0][1])
    info = m.info()
    assert info["exchanges"] == o[0][2] and info["rebalances"] == final[0][1] > 3 and info["grid_layout"] == 1


@pytest.mark.parametrize("exact", ["1", "0"])
def test_halo_messages_carry_what_was_packed(exact, monkeypatch):
    """Round 6: with the record counts exchanged first (SPHX_EXACT_EXCHANGE=1; the default for messages of >= 1 MiB at capacity) a halo
    message is (1 + records) * 32 bytes rounded up to 64 KiB, not the buffers' capacity — same results either way, bit for bit."""
    monkeypatch.setenv("SPHX_EXACT_EXCHANGE", exact)
    pos, boundary = dam_break(2.0)
    world, axis, halo, steps = 3, 1, 10, 40
    cuts = quantile_cuts(cell_coord(pos, axis), world)
    o, _ = oracle_tiles(pos, boundary, world, axis, steps, halo=halo, fixed=(2, 2), cuts=cuts, adaptive_halo=True, rebalance_every=0)
    m = MultiSolver(y.default_params(fixed_iterations=(2, 2)), devices=[0] * world, halo=halo, rebalance_every=0)
    m.set_strips(axis, cuts)
    m.set_boundary(boundary)
    m.upload(pos)
    timer = y.TimeManager()
    stats = [m.step(timer) for _ in range(steps)]
    compare(m.download(), o, stats, o[0][1])
    info = m.info()
    ex, peers, cap_bytes = info["exchanges"], info["peers"], (1 + info["cap_records"]) * 32
    assert ex >= steps and peers >= 1
    if exact == "1":
        assert 0 < info["halo_bytes_packed"] <= info["halo_bytes_sent"] <= ex * peers * cap_bytes
        assert info["halo_bytes_sent"] - info["halo_bytes_packed"] < ex * peers * 65536  # rounded up to 64 KiB per message
        assert info["halo_bytes_sent"] < 0.9 * ex * peers * cap_bytes  # the capacity is 1.5 x the set-up's estimate + 1024 records
    else:
        assert info["halo_bytes_packed"] == 0 and info["halo_bytes_sent"] <= ex * peers * cap_bytes
    assert info["ownership_seconds"] > 0.0


def test_automatic_layout_is_the_quantile_layout():
    """No cuts given: strips along the longer side at particle-count quantiles (2x2 on four tiles), like bench.py chose them."""
    pos, boundary = dam_break(1.5)
    for world in (2, 4):
        m = MultiSolver(y.default_params(), devices=[0] * world, halo=8)
        m.set_boundary(boundary)
        m.upload(pos)
        timer = y.TimeManager()
        for _ in range(5):
            m.step(timer)
        d = m.download()
        assert np.array_equal(np.sort(d["ids"]), np.arange(len(pos), dtype=np.uint32))
        assert m.info()["grid_layout"] == (1 if world == 4 else 0)
        m.close()


def test_one_process_per_tile_over_a_caller_supplied_communicator(tmp_path):
    """Two processes share the GPU; the library's step loop calls back into torch.distributed (gloo) for the halo records.  Same bits
    as the two tiles held in one process."""
    steps, scale = 60, 1.0
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(HERE, "multi_rank_worker.py"), str(tmp_path), str(steps), str(scale), "0", "0"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    ranks = by_id({k: np.concatenate([r[0][k], r[1][k]]) for k in ("ids", "pos", "vel", "density")})
    pos, boundary = dam_break(scale)
    m = MultiSolver(y.default_params(), devices=[0, 0], halo=10, rebalance_every=4)
    m.set_boundary(boundary)
    m.upload(pos)
    timer = y.TimeManager()
    stats = [m.step(timer) for _ in range(steps)]
    local = by_id(m.download())
    np.testing.assert_array_equal(ranks["ids"], local["ids"])
    for k in ("pos", "vel", "density"):
        assert_bits_equal(ranks[k], local[k], k)
    assert [s["dt_ns"] for s in stats] == r[0]["dt_ns"].tolist() == r[1]["dt_ns"].tolist()
    assert int(r[0]["exchanges"]) == m.info()["exchanges"]


def test_builtin_rccl_transport_comes_up(monkeypatch):
    """sphx_multi_create_rank without a communicator: shared-memory segment + RCCL loaded at run time (dlopen, ncclGetUniqueId,
    ncclCommInitRank).  One rank is all this box allows; the step loop then runs through the same code path with no peers."""
    monkeypatch.setenv("SPHX_RCCL_ALWAYS", "1")
    pos, boundary = dam_break(1.0)
    m = MultiSolver.rank(y.default_params(), 0, 0, 1, comm=None, job="pytest-rccl")
    assert "RCCL" in m.info()["transport"]
    m.set_boundary(boundary)
    m.upload(pos)
    timer = y.TimeManager()
    for _ in range(20):
        m.step(timer)
    d = m.download()
    assert len(d["ids"]) == len(pos) and np.isfinite(d["pos"]).all()
    # one tile, no cuts: the same particles as the single context until the first warm start (tile mode lets kappa travel)
    ctx = y.SphxContext()
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    t2 = y.TimeManager()
    for _ in range(20):
        vmax = ctx.step_begin(t2.simulation_step())
        ctx.step_finish(y.duration_as_secs_f32(t2.update_simulation_step(np.float32(0.01), vmax)))
    s = ctx.download()
    a, b = by_id(d), by_id(dict(ids=s["ids"], pos=s["pos"], vel=s["vel"], density=s["density"]))
    for k in ("pos", "vel", "density"):
        assert_bits_equal(a[k], b[k], k)


def test_errors_are_reported_not_thrown():
    L = _lib.lib()
    assert L.sphx_multi_create(None, None, 0, None, None) == _lib.ERR_INVALID_ARGUMENT
    m = MultiSolver(y.default_params(), devices=[0, 0], halo=16)
    pos, boundary = dam_break(1.0)
    m.set_boundary(boundary)
    with pytest.raises(y.SphxError):
        m.set_strips(0, [0, 10, 20, 65536])  # three tiles for a two-tile solver
    with pytest.raises(y.SphxError):
        m.step_finish(0.001)  # no step open
    m.set_strips(1, [0, 5060, 65536])
    m.upload(pos)
    with pytest.raises(y.SphxError):
        m.step_finish(0.001)
    v = m.step_begin(1.0 / 360.0)
    with pytest.raises(y.SphxError):
        m.step_begin(1.0 / 360.0)  # twice
    with pytest.raises(y.SphxError):
        m.step_finish(float("nan"))
    assert v >= 0.0


def test_solver_trait_over_a_device_list():
    """sph::HipDfsphMultiSolver: the caller keeps ONE Box<dyn Solver> and its one simulation_step(&mut world, &mut time_manager) per
    step (main.rs:50, :279); the device list is an argument of the constructor.  Same bits as driving sphx_multi directly; the world
    can be synced and edited mid-run like with the single-GPU solver."""
    w = y.FluidParticleWorld()
    w.reset_fluid(1.0)
    pos, boundary = w.positions, w.boundary_particles
    t = y.TimeManager()
    s = y.DFSPHMultiSolver(w, devices=[0, 0])
    stats = [s.simulation_step(w, t, sync_world=False) for _ in range(40)]
    s.sync_world(w)
    a = by_id(dict(ids=w.particle_ids, pos=w.positions, vel=w.velocities, density=w.densities))
    m = MultiSolver(y.default_params(), devices=[0, 0])
    m.set_boundary(boundary)
    m.upload(pos)
    t2 = y.TimeManager()
    ref = m.steps(t2, 15) + [m.step(t2) for _ in range(10)] + m.steps(t2, 15)  # the frame-loop call = single calls
    b = by_id(m.download())
    np.testing.assert_array_equal(a["ids"], b["ids"])
    for k in ("pos", "vel", "density"):
        assert_bits_equal(a[k], b[k], k)
    assert [x["divergence_iterations"] for x in stats] == [x["divergence_iterations"] for x in ref]
    assert t.simulation_step_ns() == t2.simulation_step_ns()
    # the caller edits the (synced) world: a fresh decomposition, the run goes on
    w.add_fluid_rect(1.2, 1.0, 0.2, 0.2, 0.05)
    n2 = w.num_dynamic_particles
    for _ in range(10):
        s.simulation_step(w, t, sync_world=True)
    assert w.num_dynamic_particles == n2 > len(pos) and np.isfinite(w.positions).all()


def test_regrid_with_a_folded_divergence_pass_keeps_its_contract():
    """sphx_sub_regrid_div / sphx_sub_regrid_warm: the re-grid's neighbour build does the first pass of the divergence loop that
    follows.  What follows must be that loop: after _div the first divergence iteration (no warm start: the pass has zeroed the
    warm-start values), after _warm the divergence warm start (a no-op then).  Anything else is refused, nothing is launched."""
    L = _lib.lib()
    m = MultiSolver(y.default_params(), devices=[0], halo=16)
    pos, boundary = dam_break(1.0)
    m.set_boundary(boundary)
    m.upload(pos)
    t = y.TimeManager()
    for _ in range(3):
        m.step(t)
    ctx = C.c_void_p(L.sphx_multi_tile_ctx(m.h, 0))
    n = C.c_uint32()
    s, owned = C.c_double(), C.c_uint64()
    dt = np.float32(0.002)
    assert L.sphx_sub_regrid_div(ctx, C.byref(n)) == 0
    assert L.sphx_sub_warmstart(ctx, 1, dt) == _lib.ERR_NOT_READY          # a warm start cannot follow
    assert L.sphx_sub_iteration(ctx, 0, dt, 1, C.byref(s), C.byref(owned)) == _lib.ERR_NOT_READY  # nor a density iteration
    assert L.sphx_sub_regrid_div(ctx, C.byref(n)) == 0
    assert L.sphx_sub_iteration(ctx, 1, dt, 1, C.byref(s), C.byref(owned)) == 0 and owned.value == n.value == len(pos)
    a = s.value
    assert L.sphx_sub_regrid(ctx, C.byref(n)) == 0
    assert L.sphx_sub_iteration(ctx, 1, dt, 1, C.byref(s), C.byref(owned)) == 0
    assert s.value >= 0 and np.isfinite(a)
    assert L.sphx_sub_regrid_warm(ctx, C.byref(n)) == 0
    assert L.sphx_sub_warmstart(ctx, 0, dt) == _lib.ERR_NOT_READY           # the DIVERGENCE loop's warm start was applied
    assert L.sphx_sub_regrid_warm(ctx, C.byref(n)) == 0
    assert L.sphx_sub_warmstart(ctx, 1, dt) == 0                            # accepted, nothing launched
    assert L.sphx_sub_iteration(ctx, 1, dt, 1, C.byref(s), C.byref(owned)) == 0


def test_multi_solver_keeps_particles_appended_behind_a_headless_world():
    """Round-2 advisor finding: HipDfsphMultiSolver::simulation_step fetched the stale prefix with sync_world(), which shrank the host
    arrays to the owned particles — the particles the caller had appended since (add_fluid_rect after headless steps) were dropped
    and the upload read past the vectors.  Headless and synced stepping must see the same scene: same particle count, same
    iteration counts and time steps, same state particle for particle (the two runs cut the same tiles from the same arrays)."""

    def run(sync):
        w = y.FluidParticleWorld()
        w.reset_fluid(1.0)
        t = y.TimeManager()
        s = y.DFSPHMultiSolver(w, devices=[0, 0])
        st = [s.simulation_step(w, t, sync_world=sync) for _ in range(25)]
        w.add_fluid_rect(1.2, 1.0, 0.2, 0.2, 0.05)  # appended behind the particles the tiles are working on
        n_after_edit = w.num_dynamic_particles
        st += [s.simulation_step(w, t, sync_world=sync) for _ in range(10)]
        s.sync_world(w)
        return n_after_edit, w.num_dynamic_particles, np.array(w.positions), np.array(w.velocities), t.total_simulated_ns, [x["divergence_iterations"] for x in st]

    ea, na, pa, va, ta, ia = run(True)
    eb, nb, pb, vb, tb, ib = run(False)
    assert na == nb == ea > 4050, (ea, eb, na, nb)
    assert ta == tb and ia == ib
    # the host order after a sync is tile after tile; compare as sets of (position, velocity) records
    ra = np.sort(np.ascontiguousarray(np.concatenate([pa, va], axis=1)).view([("", np.float32)] * 4).ravel())
    rb = np.sort(np.ascontiguousarray(np.concatenate([pb, vb], axis=1)).view([("", np.float32)] * 4).ravel())
    assert np.array_equal(ra, rb)


def test_sphx_multi_four_tiles_at_4M_against_the_single_context():
    """The C++ tile loop itself (sphx_multi, not its Python reference) at a size where things happen: 4 M particles on 2 x 2 tiles
    of ONE device, started from a deliberately skewed cut (30 % / 70 % columns) so that the re-partitioning has work to do, with the
    adaptive ghost band.  Against the single context on the same scene, by particle id: bit-equal after the first step (no
    migration yet), summation-order accuracy after 24 steps; every particle owned exactly once; cuts moved, band narrowed; the
    measured list statistics of the tiles (sphx_multi_info) agree with the single context's."""
    pos, boundary = dam_break(float(np.sqrt(4.0e6 / 4050.0)))
    n = len(pos)
    assert n > 3_900_000
    steps = 24
    # single context
    ctx = y.SphxContext()
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    timer = y.TimeManager()
    single = {}
    for k in range(1, steps + 1):
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(np.float32(0.01)))
        st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(np.float32(0.01), vmax)))
        if k in (1, steps):
            single[k] = (by_id({kk: vv for kk, vv in ctx.download(density=False).items() if kk in ("ids", "pos", "vel")}), st, timer.simulation_step_ns())
    kbar_single = st["neighbor_entries"] / n
    ctx.close()
    # four tiles, skewed start
    cx, cy = cell_coord(pos, 0), cell_coord(pos, 1)
    xs = np.sort(cx)
    xcuts = [0, int(xs[int(0.3 * n)]), 65536]
    ycuts = []
    for ix in range(2):
        col = cy[(cx >= xcuts[ix]) & (cx < xcuts[ix + 1])]
        ycuts.append([0, int(np.sort(col)[len(col) // 2]), 65536])
    m = MultiSolver(y.default_params(), devices=[0, 0, 0, 0], rebalance_every=4)
    m.set_grid(xcuts, np.array(ycuts, np.uint32))
    m.set_boundary(boundary)
    m.upload(pos)
    t2 = y.TimeManager()
    st1 = m.step(t2)
    a1 = by_id({k: v for k, v in m.download().items() if k in ("ids", "pos", "vel")})
    ref1, rst1, rdt1 = single[1]
    np.testing.assert_array_equal(a1["ids"], np.arange(n, dtype=np.uint32))  # every particle owned exactly once
    assert_bits_equal(a1["pos"], ref1["pos"], "pos after one step")
    assert_bits_equal(a1["vel"], ref1["vel"], "vel after one step")
    assert (st1["density_iterations"], st1["divergence_iterations"], st1["dt_ns"]) == (rst1["density_iterations"], rst1["divergence_iterations"], rdt1)
    stats = m.steps(t2, steps - 1)
    ak = by_id({k: v for k, v in m.download().items() if k in ("ids", "pos", "vel")})
    refk, rstk, rdtk = single[steps]
    np.testing.assert_array_equal(ak["ids"], np.arange(n, dtype=np.uint32))
    np.testing.assert_allclose(ak["pos"], refk["pos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ak["vel"], refk["vel"], rtol=1e-4, atol=1e-5)
    assert t2.simulation_step_ns() == rdtk and stats[-1]["divergence_iterations"] == rstk["divergence_iterations"]
    info = m.info()
    assert info["rebalances"] >= 3, info          # the skewed cuts were moved
    assert info["halo_now"] < info["halo_max"], info  # Id = Iv = 1: the band in use is narrower than the widest one
    assert info["exchanges"] >= steps + 1, info
    assert info["owned_local"] == n and info["build_particles"] > n  # owned + ghosts
    kbar_tiles = info["neighbor_entries"] / info["build_particles"]
    assert abs(kbar_tiles - kbar_single) < 0.15, (kbar_tiles, kbar_single)  # (ghosts at the rim of the band have shorter lists)
    assert 0 < info["remote_entries"] < info["neighbor_entries"]
    m.close()


def test_sphx_multi_equals_the_single_context_bit_for_bit_in_tiling_invariant_mode():
    """The C++ tile loop (sphx_multi) on 2 x 2 tiles of ONE device against the single context, both in tiling-invariant mode
    (sphx_set_tiling_invariant: cell mates ordered by persistent id, warm-start values travel).  4 M particles from a skewed cut so that
    particles migrate and the cuts move; fixed 2 + 2 iterations per step so that BOTH warm starts fire on every step from the second on
    (the adaptive run of this scene needs none in its first steps).  Bit-equal positions and velocities after 12 steps, every particle
    owned exactly once."""
    pos, boundary = dam_break(float(np.sqrt(4.0e6 / 4050.0)))
    n, steps = len(pos), 12
    params = y.default_params(fixed_iterations=(2, 2))
    ctx = y.SphxContext(params)
    ctx.set_tiling_invariant(True)
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    timer = y.TimeManager()
    warm = 0
    for _ in range(steps):
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(np.float32(0.01)))
        st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(np.float32(0.01), vmax)))
        warm += st["warmstart_divergence"] + st["warmstart_density"]
    assert warm >= 2 * (steps - 1)
    ref = by_id({kk: vv for kk, vv in ctx.download(density=False).items() if kk in ("ids", "pos", "vel")})
    ctx.close()
    cx, cy = cell_coord(pos, 0), cell_coord(pos, 1)
    xs = np.sort(cx)
    xcuts = [0, int(xs[int(0.3 * n)]), 65536]
    ycuts = []
    for ix in range(2):
        col = cy[(cx >= xcuts[ix]) & (cx < xcuts[ix + 1])]
        ycuts.append([0, int(np.sort(col)[len(col) // 2]), 65536])
    m = MultiSolver(y.default_params(fixed_iterations=(2, 2)), devices=[0, 0, 0, 0], rebalance_every=4)
    m.set_tiling_invariant(True)
    m.set_grid(xcuts, np.array(ycuts, np.uint32))
    m.set_boundary(boundary)
    m.upload(pos)
    t2 = y.TimeManager()
    m.steps(t2, steps)
    a = by_id({k: v for k, v in m.download().items() if k in ("ids", "pos", "vel")})
    np.testing.assert_array_equal(a["ids"], np.arange(n, dtype=np.uint32))
    assert_bits_equal(a["pos"], ref["pos"], "positions, tiling-invariant mode")
    assert_bits_equal(a["vel"], ref["vel"], "velocities, tiling-invariant mode")
    assert t2.simulation_step_ns() == timer.simulation_step_ns()
    assert m.info()["rebalances"] >= 2
    m.close()


def _sphx_multi_vs_single_in_tiling_invariant_mode(target, steps, make_multi):
    """The loop `bench.py --gpus N` runs (sphx_multi, csrc/sphx_tiles.cpp) at a BASELINE config's full size on ONE device, against the
    single context, both in tiling-invariant mode with fixed 2 + 2 iterations (both warm starts fire from the second step on):
    bit-equal positions and velocities by particle id, every particle owned exactly once, the same timer, >= 1 re-partition."""
    import gc

    pos, boundary = dam_break(float(np.sqrt(target / 4050.0)))
    n = len(pos)
    assert 0.97 * target < n < 1.03 * target
    params = y.default_params(fixed_iterations=(2, 2))
    ctx = y.SphxContext(params)
    ctx.set_tiling_invariant(True)
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    timer = y.TimeManager()
    warm = 0
    for _ in range(steps):
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(np.float32(0.01)))
        st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(np.float32(0.01), vmax)))
        warm += st["warmstart_divergence"] + st["warmstart_density"]
    assert warm >= 2 * (steps - 1)
    ref = by_id({kk: vv for kk, vv in ctx.download(density=False).items() if kk in ("ids", "pos", "vel")})
    ctx.close()
    del ctx
    gc.collect()
    m = make_multi(pos, y.default_params(fixed_iterations=(2, 2)))
    m.set_tiling_invariant(True)
    m.set_boundary(boundary)
    m.upload(pos)
    t2 = y.TimeManager()
    m.steps(t2, steps)
    info = m.info()
    a = by_id({k: v for k, v in m.download().items() if k in ("ids", "pos", "vel")})
    m.close()
    assert len(a["ids"]) == n and (a["ids"] == np.arange(n, dtype=np.uint32)).all()  # every particle owned exactly once
    assert_bits_equal(a["pos"], ref["pos"], "positions, tiling-invariant mode")
    assert_bits_equal(a["vel"], ref["vel"], "velocities, tiling-invariant mode")
    assert t2.simulation_step_ns() == timer.simulation_step_ns()
    assert info["rebalances"] >= 1, info
    assert info["exchanges"] >= steps + 1, info
    assert info["owned_local"] == n
    return info


def test_sphx_multi_2x2_tiles_at_64M_equal_the_single_context_bit_for_bit():
    """BASELINE configs[3] (DFSPH 64 M particles, 2 x 2 spatial tiles) through the C++ tile loop the bench uses, four tile contexts on
    one MI355X; skewed cuts (30 % / 70 % columns) so that the first re-partition moves them."""

    def make(pos, params):
        cx, cy = cell_coord(pos, 0), cell_coord(pos, 1)
        n = len(pos)
        xcuts = [0, int(np.partition(cx, int(0.3 * n))[int(0.3 * n)]), 65536]
        ycuts = []
        for ix in range(2):
            col = cy[(cx >= xcuts[ix]) & (cx < xcuts[ix + 1])]
            ycuts.append([0, int(np.partition(col, len(col) // 2)[len(col) // 2]), 65536])
        m = MultiSolver(params, devices=[0, 0, 0, 0], rebalance_every=2)
        m.set_grid(xcuts, np.array(ycuts, np.uint32))
        return m

    _sphx_multi_vs_single_in_tiling_invariant_mode(64.0e6, 4, make)


def test_sphx_multi_8_strips_at_128M_equal_the_single_context_bit_for_bit():
    """BASELINE configs[4] (DFSPH 128 M particles, 8 strips) through the C++ tile loop the bench uses, eight tile contexts on one
    MI355X; the first strip starts 40 % too wide so that the first re-partition moves the cuts."""

    def make(pos, params):
        ext = pos.max(0) - pos.min(0)
        axis = int(ext[1] > ext[0])
        c = cell_coord(pos, axis)
        n = len(pos)
        fr = [0.175] + [0.175 + (1.0 - 0.175) * k / 7.0 for k in range(1, 7)]
        inner = [int(np.partition(c, int(f * n))[int(f * n)]) for f in fr]
        cuts = [0] + inner + [65536]
        assert all(b > a for a, b in zip(cuts[:-1], cuts[1:]))
        m = MultiSolver(params, devices=[0] * 8, rebalance_every=2)
        m.set_strips(axis, cuts)
        return m

    _sphx_multi_vs_single_in_tiling_invariant_mode(128.0e6, 3, make)


@pytest.mark.parametrize("overlap", [False, True])
def test_tile_classification_by_the_last_density_correction_changes_nothing(monkeypatch, overlap):
    """The density loop's last correction holds the advected positions: in a tile it counts the cells of the particles the tile keeps
    and the records to send (TileClassArgs), the packing pass then visits only the workgroups inside a send band, the re-grid's gather
    clears the owner bit of new ghosts.  SPHX_TILE_FUSE_CLASS=0: k_tile_count + k_tile_pack read every particle again.  2 x 2 tiles
    through the impact with adaptive iteration counts (warm starts, extra exchanges in the middle of a loop, re-partitioning, band
    adaptation — each of which makes single steps fall back): same bits, counts, exchanges; and the fused path was the usual one."""
    pos, boundary = dam_break(2.0)
    lay = GridLayout.quantile(pos, 2, 2)

    def run(fused):
        monkeypatch.setenv("SPHX_TILE_FUSE_CLASS", "1" if fused else "0")
        m = MultiSolver(y.default_params(), devices=[0, 0, 0, 0], halo=10, rebalance_every=6, overlap_exchange=overlap)
        m.set_grid(lay.xcuts, lay.ycuts)
        m.set_boundary(boundary)
        m.upload(pos)
        timer = y.TimeManager()
        stats = [m.step(timer) for _ in range(140)]
        out = by_id({k: v for k, v in m.download().items() if k in ("ids", "pos", "vel", "density")})
        return out, [(s["density_iterations"], s["divergence_iterations"], s["dt_ns"]) for s in stats], m.info()

    a, ca, ia = run(True)
    b, cb, ib = run(False)
    assert ca == cb and ia["exchanges"] == ib["exchanges"] and ia["rebalances"] == ib["rebalances"]
    np.testing.assert_array_equal(a["ids"], b["ids"])
    for k in ("pos", "vel", "density"):
        assert_bits_equal(a[k], b[k], k)
    assert ib["band_packs"] == 0
    assert ia["band_packs"] > 100, ia  # (140 steps; the steps with a re-partitioning or a new band width pack the long way)


def test_classified_tile_step_survives_a_call_between_pack_and_regrid():
    """Sub-step ABI: after a pack that used the density correction's classification the advection of the kept records, the owner
    bits of the new ghosts and the retirement of what left the band are all pending in the re-grid's gather.  A caller that looks at
    the records first (here: a download) must find them applied, and the re-grid behind it must end in the same state."""
    import torch

    L = _lib.lib()
    pos, boundary = dam_break(1.0)
    cx = cell_coord(pos, 0)
    cut = int(np.sort(cx)[len(cx) // 2])
    rect = C.c_uint32 * 4
    mine = np.nonzero(cx < cut)[0]
    p = np.ascontiguousarray(pos[mine], np.float32)
    v = np.zeros_like(p)
    v[:, 0] = 3.0     # everything drifts towards the cut: owned particles cross it and stay as ghosts ...
    v[::53, 0] = 90.0  # ... and some jump over the whole ghost band: retired
    ids = mine.astype(np.uint32)
    dt = np.float32(0.004)
    cap = 20000
    res = {}
    for mode in ("regrid", "flush"):
        ctx = y.SphxContext()
        h = ctx.h
        ctx.set_boundary(boundary)
        assert L.sphx_tile_configure_rect(h, rect(0, cut, 0, 65536), 8, (rect * 1)(rect(cut, 65536, 0, 65536)), 1) == 0
        assert L.sphx_tile_defer_advect(h, 1) == 0
        assert L.sphx_reserve(h, len(mine) + cap) == 0
        q = lambda a: a.ctypes.data_as(C.c_void_p)
        assert L.sphx_tile_upload(h, q(p), q(v), q(ids), len(mine)) == 0
        n = C.c_uint32()
        assert L.sphx_sub_regrid(h, C.byref(n)) == 0 and n.value == len(mine)
        vsq = C.c_float()
        assert L.sphx_sub_nonpressure(h, dt, C.byref(vsq)) == 0
        s_, owned = C.c_double(), C.c_uint64()
        assert L.sphx_sub_predict_iteration(h, dt, C.byref(s_), C.byref(owned)) == 0 and owned.value == len(mine)
        send = torch.zeros((1 + cap) * 32, dtype=torch.uint8, device="cuda")
        bufs = (C.c_void_p * 1)(send.data_ptr())
        assert L.sphx_tile_advect_pack_n(h, dt, bufs, 1, cap) == 0
        packs = C.c_uint32()
        assert L.sphx_tile_band_packs(h, C.byref(packs)) == 0 and packs.value == 1
        if mode == "flush":
            res["flush_pos"] = ctx.download(density=False)["pos"].copy()  # looks at the records: the pending work is applied first
        assert L.sphx_sub_regrid(h, C.byref(n)) == 0
        d = ctx.download(density=False)
        res[mode] = (n.value, {k: d[k][np.argsort(d["ids"] & 0x7FFFFFFF, kind="stable")] for k in ("ids", "pos", "vel")}, send.cpu().numpy().copy())
        ctx.close()
    na, a, sa = res["regrid"]
    nb, b, sb = res["flush"]
    assert na == nb < len(mine) and np.array_equal(sa, sb)
    np.testing.assert_array_equal(a["ids"], b["ids"])  # (owner bits included)
    assert_bits_equal(a["pos"], b["pos"], "pos")
    assert_bits_equal(a["vel"], b["vel"], "vel")
    assert 0 < (a["ids"] >> 31).sum() < na            # some owned particles crossed the cut and stayed as ghosts
    assert np.isnan(res["flush_pos"][:, 0]).sum() == len(mine) - na  # the ones that left the band: retired by the flush already
    assert int(np.frombuffer(sa[16:20].tobytes(), np.uint32)[0]) > 0   # header of the send buffer: records went out
