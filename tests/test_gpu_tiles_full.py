"""BASELINE configs[3] (64 M particles, 2x2 tiles) and configs[4] (128 M particles, 8 strips) through the TILE path at full size,
on one MI355X: the tiles run as threads of this process, one libsphx context each (288 GB of HBM hold both runs), halo records go
through device buffers exactly as they would through RCCL.

What is compared with the single-context run of the same scene ("Parity across tilings"; reference semantics:
neighborhood_search.rs:348-393), by persistent particle id:

  after ONE step   every particle is owned by exactly one tile; positions and velocities BIT-equal; cell membership (hence the
                   per-cell counts of the Morton grid) identical; per-particle neighbour counts and neighbour ID SETS identical
                   (dynamic neighbours by particle id, static ones by boundary position) — on every particle at 4 M, on a
                   sample of 60 000 particles (a third of them next to the cuts) at 64 M / 128 M, where hashing a billion list
                   entries on the host is not worth the minutes;
  after k steps    positions 1e-5, velocities 1e-4 relative.  Bit-equality ends with the first migration: a particle that crosses
                   a cut arrives at the END of its new tile's array, so inside its new cell it sorts (stable sort by previous
                   index) behind its cell mates, whereas the single context keeps it where its previous sorted index puts it —
                   the neighbour SETS stay equal, the order of the per-particle sums and thus the last bits do not.

Both dependencies on the tiling — that order, and the warm-start values, which tile mode lets travel with the particle —
go away in tiling-invariant mode (sphx_set_tiling_invariant): test_tiles_equal_the_single_context_through_warm_starts_in_tiling_invariant_mode
compares tiles and single context bit for bit through the impact.
"""
import threading

import numpy as np
import pytest
from util import dam_break

import yasph2d_amd as y
from tiles_reference import GpuTileBackend, GridLayout, StripLayout, ThreadComm, TiledDFSPH, cell_coord, quantile_cuts

pytestmark = pytest.mark.gpu

M64 = np.uint64(0x9E3779B97F4A7C15)


def mix(a):
    """64-bit mixer (SplitMix64 finaliser) for order-independent set hashes."""
    a = a.astype(np.uint64)
    a = (a ^ (a >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    a = (a ^ (a >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return a ^ (a >> np.uint64(31))


def neighbour_hashes(ctx, ids_local, sample_local):
    """For the local particles `sample_local`: (count_dynamic, count_total, hash of the dynamic neighbour ID set, hash of the static
    neighbour position set)."""
    counts, start, lists = ctx.download_neighbors()
    bxy, _ = ctx.download_boundary()
    bkey = mix(bxy[:, 0].view(np.uint32).astype(np.uint64) << np.uint64(32) | bxy[:, 1].view(np.uint32).astype(np.uint64)) if len(bxy) else np.zeros(0, np.uint64)
    s = sample_local.astype(np.int64)
    ct = counts[s, 1].astype(np.int64)
    cd = counts[s, 0].astype(np.int64)
    owner = np.repeat(np.arange(len(s)), ct)
    k = np.arange(int(ct.sum())) - np.repeat(np.cumsum(ct) - ct, ct)
    entry = lists[np.repeat(start[s].astype(np.int64), ct) + k].astype(np.int64)
    dyn = k < cd[owner]
    hd = np.zeros(len(s), np.uint64)
    hs = np.zeros(len(s), np.uint64)
    with np.errstate(over="ignore"):
        np.add.at(hd, owner[dyn], mix(ids_local[entry[dyn]].astype(np.uint64) + np.uint64(1)))
        if (~dyn).any():
            np.add.at(hs, owner[~dyn], bkey[entry[~dyn]])
    return cd, ct, hd, hs


def cell_keys(pos):
    return (cell_coord(pos, 1).astype(np.uint64) << np.uint64(16)) | cell_coord(pos, 0).astype(np.uint64)


def run_single(pos, boundary, steps_list, sample_ids):
    """-> {steps: dict(pos, vel by id [+ neighbour hashes of sample_ids after the first entry])}"""
    ctx = y.SphxContext()
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    timer = y.TimeManager()
    out, done = {}, 0
    for k in steps_list:
        for _ in range(k - done):
            vmax = ctx.step_begin(timer.simulation_step(), timer.law(np.float32(0.01)))
            st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(np.float32(0.01), vmax)))
        done = k
        d = ctx.download(density=False)
        inv = np.argsort(d["ids"])
        rec = dict(pos=d["pos"][inv], vel=d["vel"][inv], stats=st, dt_ns=timer.simulation_step_ns())
        if not out:
            rank_of = np.empty(len(inv), np.int64)
            rank_of[d["ids"]] = np.arange(len(inv))
            rec["nb"] = neighbour_hashes(ctx, d["ids"], rank_of[sample_ids])
        out[k] = rec
    ctx.close()
    return out


def run_tiles(pos, boundary, world, layout_factory, steps_list, sample_ids, halo=16, cap=None, invariant=False, keep_stats=False):
    shared = ThreadComm.Shared(world)
    res, errs = [None] * world, []
    n = len(pos)

    def work(r):
        try:
            ctx = y.SphxContext()
            if invariant:
                ctx.set_tiling_invariant(True)
            t = TiledDFSPH(GpuTileBackend(ctx), ThreadComm(shared, r), layout_factory(), halo=halo, adaptive_halo=True, cap_records=cap)
            t.setup(pos, None, None, boundary)
            timer = y.TimeManager()
            out, done = {}, 0
            all_stats = []
            for k in steps_list:
                for _ in range(k - done):
                    st = t.step(timer)
                    if keep_stats:
                        all_stats.append(dict(st, dt_ns=timer.simulation_step_ns()))
                done = k
                d = t.b.download()
                own = d["owned"]
                rec = dict(ids=d["ids"][own], pos=d["pos"][own], vel=d["vel"][own], stats=st, exchanges=t.exchanges)
                if not out:
                    # sampled particles this tile owns: local index of each
                    loc = np.full(n, -1, np.int64)
                    loc[d["ids"][own]] = np.nonzero(own)[0]
                    mine = loc[sample_ids] >= 0
                    rec["nb_sample"] = np.nonzero(mine)[0]
                    rec["nb"] = neighbour_hashes(ctx, d["ids"], loc[sample_ids[mine]])
                out[k] = rec
            if keep_stats:
                out["all_stats"] = all_stats
            res[r] = out
            ctx.close()
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    return res


def merge(res, k, n):
    p, v = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32)
    seen = np.zeros(n, np.uint8)
    for out in res:
        r = out[k]
        p[r["ids"]], v[r["ids"]] = r["pos"], r["vel"]
        seen[r["ids"]] += 1
    assert (seen == 1).all(), "every particle must be owned by exactly one tile"
    return p, v


def sample_near_cuts(pos, rects, n_sample, seed):
    """Two thirds random particles, one third within two cells of a tile edge (where a wrong halo would show)."""
    rng = np.random.default_rng(seed)
    cx, cy = cell_coord(pos, 0).astype(np.int64), cell_coord(pos, 1).astype(np.int64)
    near = np.zeros(len(pos), bool)
    for x0, x1, y0, y1 in rects:
        for c, lo, hi in ((cx, x0, x1), (cy, y0, y1)):
            for edge in (lo, hi):
                if 0 < edge < 65536:
                    near |= np.abs(c - edge) <= 2
    near_idx = np.nonzero(near)[0]
    a = rng.choice(near_idx, min(len(near_idx), n_sample // 3), replace=False) if len(near_idx) else np.zeros(0, np.int64)
    b = rng.integers(0, len(pos), n_sample - len(a))
    return np.unique(np.concatenate([a, b])).astype(np.int64)


def check_against_single(pos, boundary, world, layout_factory, steps, sample_ids, cap):
    n = len(pos)
    single = run_single(pos, boundary, [1, steps], sample_ids)
    tiles = run_tiles(pos, boundary, world, layout_factory, [1, steps], sample_ids, cap=cap)
    # ---- after one step: bit-equal state, identical cells, identical neighbour sets
    p1, v1 = merge(tiles, 1, n)
    assert np.array_equal(p1.view(np.uint32), single[1]["pos"].view(np.uint32)), "positions after one step must be bit-equal"
    assert np.array_equal(v1.view(np.uint32), single[1]["vel"].view(np.uint32)), "velocities after one step must be bit-equal"
    assert np.array_equal(np.sort(cell_keys(p1)), np.sort(cell_keys(single[1]["pos"])))  # same multiset of cells = same per-cell counts
    for out in tiles:
        st = out[1]["stats"]
        assert st["dt_ns"] == single[1]["dt_ns"]
        assert st["density_iterations"] == single[1]["stats"]["density_iterations"]
        assert st["divergence_iterations"] == single[1]["stats"]["divergence_iterations"]
    cd0, ct0, hd0, hs0 = single[1]["nb"]
    covered = np.zeros(len(sample_ids), bool)
    for out in tiles:
        sel = out[1]["nb_sample"]
        cd, ct, hd, hs = out[1]["nb"]
        assert np.array_equal(cd, cd0[sel]) and np.array_equal(ct, ct0[sel]), "neighbour counts differ between tiling and single context"
        assert np.array_equal(hd, hd0[sel]), "dynamic neighbour ID sets differ between tiling and single context"
        assert np.array_equal(hs, hs0[sel]), "static neighbour sets differ between tiling and single context"
        covered[sel] = True
    assert covered.all() and ct0.sum() > 4 * len(sample_ids)
    # ---- after `steps` steps: summation-order accuracy
    pk, vk = merge(tiles, steps, n)
    np.testing.assert_allclose(pk, single[steps]["pos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vk, single[steps]["vel"], rtol=1e-4, atol=1e-5)
    assert np.isfinite(pk).all() and np.isfinite(vk).all()
    return tiles


def strips_for(pos, world):
    ext = pos.max(0) - pos.min(0)
    axis = int(ext[1] > ext[0])
    cuts = quantile_cuts(cell_coord(pos, axis), world)
    return axis, cuts


def halo_cap(pos, world_edge_particles):
    """Records per halo buffer: the particles of a 16-cell band along the longest tile edge, with head-room."""
    return int(world_edge_particles * 16 * 4 * 1.5) + 4096


@pytest.mark.parametrize("world,kind", [(4, "grid"), (8, "strips")])
def test_tilings_match_single_context_at_4M_every_particle(world, kind):
    pos, boundary = dam_break(float(np.sqrt(4.0e6 / 4050.0)))
    n = len(pos)
    if kind == "grid":
        lay = GridLayout.quantile(pos, 2, 2)
        factory = lambda: GridLayout.quantile(pos, 2, 2)  # noqa: E731
        rects = lay.rects()
    else:
        axis, cuts = strips_for(pos, world)
        factory = lambda: StripLayout(axis, cuts)  # noqa: E731
        rects = StripLayout(axis, cuts).rects()
    assert len(rects) == world
    check_against_single(pos, boundary, world, factory, 4, np.arange(n, dtype=np.int64), cap=None)


def test_config3_64M_particles_2x2_tiles():
    """BASELINE configs[3]: DFSPH 64 M particles on 2x2 spatial tiles (here: four tile contexts on one MI355X)."""
    pos, boundary = dam_break(float(np.sqrt(64.0e6 / 4050.0)))
    assert 63_000_000 < len(pos) < 65_000_000
    lay = GridLayout.quantile(pos, 2, 2)
    xc, yc = list(lay.xcuts), [list(c) for c in lay.ycuts]
    sample = sample_near_cuts(pos, lay.rects(), 60_000, 3)
    cols = max(cell_coord(pos, 0).max() - cell_coord(pos, 0).min(), cell_coord(pos, 1).max() - cell_coord(pos, 1).min())
    tiles = check_against_single(pos, boundary, 4, lambda: GridLayout(xc, yc), 3, sample, cap=halo_cap(pos, int(cols)))
    assert all(out[3]["exchanges"] >= 4 for out in tiles)  # set-up + one per step


def test_config4_128M_particles_8_strips():
    """BASELINE configs[4]: DFSPH 128 M particles on 8 strips (here: eight tile contexts on one MI355X)."""
    pos, boundary = dam_break(float(np.sqrt(128.0e6 / 4050.0)))
    assert 126_000_000 < len(pos) < 130_000_000
    axis, cuts = strips_for(pos, 8)
    sample = sample_near_cuts(pos, StripLayout(axis, cuts).rects(), 60_000, 4)
    across = cell_coord(pos, 1 - axis)
    tiles = check_against_single(pos, boundary, 8, lambda: StripLayout(axis, cuts), 2, sample, cap=halo_cap(pos, int(across.max() - across.min())))
    own = [len(out[2]["ids"]) for out in tiles]
    assert max(own) < 1.1 * (len(pos) / 8), f"quantile cuts must balance the strips: {own}"


def test_default_tile_mode_stays_physically_close_to_the_single_context_through_warm_starts():
    """The configuration bench.py --gpus N times: DEFAULT tile mode (cell mates in previous-index order, warm-start values travel)
    against the single context (slot-bound warm-start values, like dfsph.rs:512).  Once a warm start has fired the two runs are
    different — equally valid — discretisations; the tiling-invariant test below shows bit-equality in the comparison mode, this one
    keeps a long-run physical bound on the production path (round-4 advisor): through the impact of the reference scene (260 steps,
    warm starts on most of the later ones) total mechanical energy agrees to 0.5 % and the fluid's centre of mass to one spacing."""
    pos, boundary = dam_break(1.0)
    steps = 260
    ctx = y.SphxContext()
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    timer = y.TimeManager()
    warm = 0
    for _ in range(steps):
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(np.float32(0.01)))
        st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(np.float32(0.01), vmax)))
        warm += st["warmstart_divergence"] + st["warmstart_density"]
    assert warm > 30
    d = ctx.download()
    ctx.close()
    axis, cuts = 1, quantile_cuts(cell_coord(pos, 1), 2)
    tiles = run_tiles(pos, boundary, 2, lambda: StripLayout(axis, cuts), [steps], np.zeros(0, np.int64), halo=16)
    p, v = merge(tiles, steps, len(pos))

    def energy(pp, vv):
        return float((0.5 * (vv.astype(np.float64) ** 2).sum(1) + 9.81 * pp[:, 1].astype(np.float64)).sum())

    e_single, e_tiles = energy(d["pos"], d["vel"]), energy(p, v)
    assert abs(e_tiles - e_single) < 5e-3 * abs(e_single)
    assert np.abs(p.astype(np.float64).mean(0) - d["pos"].astype(np.float64).mean(0)).max() < 0.0111


def test_tiles_equal_the_single_context_through_warm_starts_in_tiling_invariant_mode():
    """Multi-GPU parity through warm starts (replaces a 0.5 % energy bound).  Tile mode lets warmstart_kappa /
    warmstart_stiffness travel with their particle (the reference leaves them slot-bound, dfsph.rs:512 — a slot means nothing across
    tiles), and a tile sorts what it receives behind what it holds, so the order inside a cell differs from the single context's.
    sphx_set_tiling_invariant removes both dependencies (cell mates ordered by persistent id, warm-start values travel).  In that
    mode, through the impact of the reference scene (260 steps, warm starts on most of the later ones):
      * the single context equals the oracle in the same mode bit for bit (the switch itself is parity-checked),
      * x-strips (the cut the collapsing column flows across) and 2 x 2 tiles equal the single context BIT FOR BIT: positions, velocities,
        iteration counts, time steps."""
    from oracle.oracle import Oracle

    pos, boundary = dam_break(1.0)
    n, steps = len(pos), 260
    o = Oracle()
    o.set_tiling_invariant(True)
    o.set_boundary(boundary)
    o.set_particles(pos)
    ctx = y.SphxContext()
    ctx.set_tiling_invariant(True)
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    timer = y.TimeManager()
    it_single, warm = [], 0
    for _ in range(steps):
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(np.float32(0.01)))
        dt_ns = timer.update_simulation_step(np.float32(0.01), vmax)
        st = ctx.step_finish(y.duration_as_secs_f32(dt_ns))
        so = o.dfsph_step()
        assert dt_ns == o.timer_step_ns()
        assert (st["density_iterations"], st["divergence_iterations"]) == (so["density_iterations"], so["divergence_iterations"])
        it_single.append((st["density_iterations"], st["divergence_iterations"], dt_ns))
        warm += st["warmstart_divergence"] + st["warmstart_density"]
    assert warm > 100
    d = ctx.download()
    inv, oinv = np.argsort(d["ids"]), np.argsort(o.ids())
    sp, sv = d["pos"][inv], d["vel"][inv]
    assert np.array_equal(sp.view(np.uint32), o.positions()[oinv].view(np.uint32)), "single context vs oracle, tiling-invariant mode: positions"
    assert np.array_equal(sv.view(np.uint32), o.velocities()[oinv].view(np.uint32)), "single context vs oracle, tiling-invariant mode: velocities"
    ss = ctx.download_solver_state()  # (the warm-start values have travelled with their particles on both sides)
    assert np.array_equal(ss["kappa"][inv].view(np.uint32), o.kappa()[oinv].view(np.uint32))
    assert np.array_equal(ss["stiffness"][inv].view(np.uint32), o.stiffness()[oinv].view(np.uint32))
    ctx.close()
    cuts = quantile_cuts(cell_coord(pos, 0), 2)
    for world, factory in ((2, lambda: StripLayout(0, cuts)), (4, lambda: GridLayout.quantile(pos, 2, 2))):
        tiles = run_tiles(pos, boundary, world, factory, [steps], np.zeros(0, np.int64), halo=16, invariant=True, keep_stats=True)
        p, v = merge(tiles, steps, n)
        for out in tiles:
            assert [(s["density_iterations"], s["divergence_iterations"], s["dt_ns"]) for s in out["all_stats"]] == it_single
        assert np.array_equal(p.view(np.uint32), sp.view(np.uint32)), f"{world} tiles vs single context: positions differ, max {np.abs(p - sp).max()}"
        assert np.array_equal(v.view(np.uint32), sv.view(np.uint32)), f"{world} tiles vs single context: velocities differ, max {np.abs(v - sv).max()}"
