"""Multi-GPU path on CPU: the tile driver (tests/tiles_reference.py — partitioning, halo records, migration, ring budget,
collectives) runs over the oracle backend with world_size-2 gloo processes and with in-process thread ranks, and must
reproduce the single-domain oracle run particle by particle (by id)."""
import os
import sys
import threading

import numpy as np
import pytest
from util import dam_break

import yasph2d_amd as y
from oracle.oracle import Oracle
from tiles_reference import GridLayout, StripLayout, ThreadComm, TiledDFSPH, cell_coord, quantile_cuts

HERE = os.path.dirname(os.path.abspath(__file__))


def single_domain(pos, boundary, steps, fixed=(0, 0)):
    o = Oracle()
    o.set_fixed_iterations(*fixed)
    o.set_boundary(boundary)
    o.set_particles(pos)
    stats = [o.dfsph_step() for _ in range(steps)]
    ids = o.ids()
    inv = np.argsort(ids)
    return dict(pos=o.positions()[inv], vel=o.velocities()[inv], density=o.densities()[inv]), stats, o.timer_step_ns()


def run_tiles_threaded(make_backend, pos, boundary, world, axis, steps, halo=16, fixed=(0, 0), cuts=None, rebalance_every=0, layout=None,
                       adaptive_halo=False):
    """axis/cuts: strips; layout: a factory returning a fresh Layout per rank (e.g. a 2x2 GridLayout)."""
    cuts = quantile_cuts(cell_coord(pos, axis), world) if cuts is None and layout is None else cuts
    shared = ThreadComm.Shared(world)
    out, errs = [None] * world, []
    final_cuts = run_tiles_threaded.final_cuts = [None] * world
    run_tiles_threaded.halos = [[] for _ in range(world)]

    def work(r):
        try:
            lay = layout() if layout is not None else StripLayout(axis, cuts)
            t = TiledDFSPH(make_backend(r), ThreadComm(shared, r), lay, halo=halo, fixed_iterations=fixed, rebalance_every=rebalance_every,
                           adaptive_halo=adaptive_halo)
            halos = run_tiles_threaded.halos[r]
            t.setup(pos, None, None, boundary)
            timer = y.TimeManager()
            stats = []
            for _ in range(steps):
                stats.append(t.step(timer))
                halos.append(t.halo_now)
            out[r] = (t.download_owned(), stats, t.exchanges)
            final_cuts[r] = (list(t.cuts), t.rebalances)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    return out, cuts


def merge_owned(outs, n):
    pos, vel, den = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32), np.zeros(n, np.float32)
    seen = np.zeros(n, np.int32)
    for d, _, _ in outs:
        pos[d["ids"]], vel[d["ids"]], den[d["ids"]] = d["pos"], d["vel"], d["density"]
        seen[d["ids"]] += 1
    assert (seen == 1).all(), "every particle must be owned by exactly one tile"
    return pos, vel, den


@pytest.mark.parametrize("world,axis,halo", [(2, 1, 16), (3, 1, 8), (2, 0, 6)])
def test_tiles_match_single_domain(world, axis, halo):
    """Adaptive timer, 60 steps (free fall, before any warm start — the slot-bound warm-start arrays of the reference
    cannot be reproduced across tiles): dt, iteration counts identical; positions/velocities to 1e-5 relative."""
    from tile_oracle_backend import OracleTileBackend

    pos, boundary = dam_break(1.0)
    steps = 60
    ref, rstats, _ = single_domain(pos, boundary, steps)
    outs, cuts = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, world, axis, steps, halo=halo)
    for s in range(steps):
        for r in range(world):
            st = outs[r][1][s]
            assert st["density_iterations"] == rstats[s]["density_iterations"] and st["divergence_iterations"] == rstats[s]["divergence_iterations"]
            assert np.float32(st["dt"]) == np.float32(rstats[s]["dt"]) and np.float32(st["vmax"]) == np.float32(rstats[s]["vmax"])
    p, v, d = merge_owned(outs, len(pos))
    np.testing.assert_allclose(p, ref["pos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v, ref["vel"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(d, ref["density"], rtol=1e-5)


def test_tiles_with_impact_migration_and_long_loops():
    """Fixed 3+2 iterations per step exercise the warm-start sub-steps and the ring budget (extra exchanges), and 200 steps
    run through the impact so particles migrate across the cut.  Warm-start values travel with the particle here, so the
    comparison is against a threaded run with ONE tile (same sub-step code path), not the slot-bound single-domain run."""
    from tile_oracle_backend import OracleTileBackend

    pos, boundary = dam_break(1.0)
    steps = 120
    one, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 1, 1, steps, fixed=(3, 2))
    two, cuts = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 2, 1, steps, fixed=(3, 2), halo=10)
    assert two[0][2] > steps + 1, "a 10-cell halo cannot cover 5 iterations + 2 warm starts per step: extra exchanges expected"
    p1, v1, _ = merge_owned(one, len(pos))
    p2, v2, _ = merge_owned(two, len(pos))
    # only the summation order inside cells differs between tilings (particles that crossed a cut sit elsewhere in their cell)
    np.testing.assert_allclose(p2, p1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v2, v1, rtol=1e-4, atol=5e-4)
    # particles changed owner
    c0 = cell_coord(pos, 1)
    c1 = cell_coord(p2, 1)
    assert ((c0 < cuts[1]) != (c1 < cuts[1])).sum() > 0


def test_rebalance_cuts_rule():
    from tiles_reference import rebalance_cuts

    cuts = [0, 5000, 5100, 5200, 65536]
    assert rebalance_cuts(cuts, [100, 100, 100, 100], 8, 10.0, 2) == cuts                     # balanced: untouched
    assert rebalance_cuts(cuts, [100, 104, 100, 100], 8, 10.0, 2) == cuts                     # below the threshold
    new = rebalance_cuts(cuts, [400, 100, 100, 100], 8, 10.0, 2)
    assert new == [0, 4998, 5100, 5200, 65536]                                                # left tile heavier: its cut moves left, clamped
    new = rebalance_cuts(cuts, [100, 100, 100, 130], 8, 10.0, 2)
    assert new == [0, 5000, 5100, 5202, 65536]                                                # 15 particles / 10 per column -> 2 columns
    tight = [0, 5000, 5018, 5036, 65536]
    new = rebalance_cuts(tight, [100, 400, 100, 100], 8, 10.0, 2)                             # tiles may not shrink below 2*halo+2 = 18
    assert all(new[r + 1] - new[r] >= 18 for r in range(1, 3))


def test_tiles_rebalance_moves_cuts_and_matches_single_domain():
    """Start from a deliberately lopsided cut (30 % / 70 %): the diffusive re-partition moves it every 2 steps while the
    simulation runs; the merged result still matches the single-domain run to summation-order accuracy."""
    from tile_oracle_backend import OracleTileBackend

    pos, boundary = dam_break(1.0)
    steps = 60
    c = cell_coord(pos, 1)
    cut = int(np.sort(c)[int(0.3 * len(c))])
    ref, rstats, _ = single_domain(pos, boundary, steps)
    outs, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 2, 1, steps, halo=8, cuts=[0, cut, 65536], rebalance_every=2)
    (cuts0, nreb0), (cuts1, nreb1) = run_tiles_threaded.final_cuts
    assert cuts0 == cuts1 and nreb0 == nreb1 > 5
    assert cuts0[1] > cut, "the cut must have moved towards the heavier tile"
    n0, n1 = len(outs[0][0]["ids"]), len(outs[1][0]["ids"])
    assert abs(n0 - n1) < 0.1 * len(pos), "the 40 % imbalance must be gone (the cut follows the falling column afterwards)"
    for s in range(steps):
        for r in range(2):
            assert outs[r][1][s]["density_iterations"] == rstats[s]["density_iterations"]
            assert np.float32(outs[r][1][s]["dt"]) == np.float32(rstats[s]["dt"])
    p, v, d = merge_owned(outs, len(pos))
    np.testing.assert_allclose(p, ref["pos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v, ref["vel"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(d, ref["density"], rtol=1e-5)


def test_adaptive_halo_follows_the_ring_budget():
    """The ghost band in use shrinks to what the solver loops spend (6-8 cells while Id = Iv = 1) and widens again when fixed 3+2
    iterations need more rings; results stay those of the single-domain run, and no step needs more than its one exchange once the
    band has adapted."""
    from tile_oracle_backend import OracleTileBackend

    pos, boundary = dam_break(1.0)
    steps = 60
    ref, rstats, _ = single_domain(pos, boundary, steps)
    outs, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 2, 1, steps, halo=16, adaptive_halo=True)
    h = run_tiles_threaded.halos
    assert h[0] == h[1] and h[0][0] <= 16 and min(h[0]) == 8 and h[0][-1] == 8, h[0]
    assert outs[0][2] == steps + 1, "one exchange per step (+ the set-up one)"
    for s in range(steps):
        assert outs[0][1][s]["density_iterations"] == rstats[s]["density_iterations"]
        assert np.float32(outs[0][1][s]["dt"]) == np.float32(rstats[s]["dt"])
    p, v, d = merge_owned(outs, len(pos))
    np.testing.assert_allclose(p, ref["pos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v, ref["vel"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(d, ref["density"], rtol=1e-5)
    # long loops: 3 + 2 fixed iterations spend 2*2+1 + 1 + 2*3+1 + 1 = 14 rings -> the band grows to 16 and stays
    one, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 1, 1, 40, fixed=(3, 2))
    two, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 2, 1, 40, fixed=(3, 2), halo=16, adaptive_halo=True)
    assert run_tiles_threaded.halos[0][-1] == 16
    p1, v1, _ = merge_owned(one, len(pos))
    p2, v2, _ = merge_owned(two, len(pos))
    np.testing.assert_allclose(p2, p1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v2, v1, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("world,axis,grid", [(2, 0, False), (4, None, True)])
def test_tiling_invariant_mode_tiles_equal_the_single_domain_bit_for_bit(world, axis, grid):
    """Multi-GPU parity THROUGH warm starts.  Two things in a run depend on the tiling: the order of the particles
    inside a cell (stable by previous index — a tile appends what it receives) and the slot-bound warm-start values of the reference
    (dfsph.rs:512; a slot means nothing across tiles).  In tiling-invariant mode (cell mates ordered by persistent id, warm-start values
    travel: Oracle.set_tiling_invariant, not the reference's behaviour) neither is left: x-strips — the cut the collapsing column flows
    across — and 2 x 2 tiles reproduce the single domain BIT FOR BIT through 260 adaptive steps of the reference scene (free fall, impact,
    splash; ~200 warm starts), iteration counts and time steps included.  (Without the mode the same comparison is exact until the first
    particle crosses a cut and drifts apart chaotically afterwards: 1e-2 by step 150.)"""
    from tile_oracle_backend import OracleTileBackend

    pos, boundary = dam_break(1.0)
    steps = 260
    o = Oracle()
    o.set_tiling_invariant(True)
    o.set_boundary(boundary)
    o.set_particles(pos)
    rstats = [o.dfsph_step() for _ in range(steps)]
    assert sum(s["warmstart_divergence"] + s["warmstart_density"] for s in rstats) > 100
    inv = np.argsort(o.ids())
    ref_p, ref_v, ref_d = o.positions()[inv], o.velocities()[inv], o.densities()[inv]

    def backend(r):
        b = OracleTileBackend()
        b.o.set_tiling_invariant(True)
        return b

    outs, _ = run_tiles_threaded(backend, pos, boundary, world, axis, steps, halo=16, layout=(lambda: GridLayout.quantile(pos, 2, 2)) if grid else None)
    for s in range(steps):
        for r in range(world):
            assert outs[r][1][s]["density_iterations"] == rstats[s]["density_iterations"], (s, r)
            assert outs[r][1][s]["divergence_iterations"] == rstats[s]["divergence_iterations"], (s, r)
    p, v, d = merge_owned(outs, len(pos))
    assert np.array_equal(p.view(np.uint32), ref_p.view(np.uint32)), f"positions differ: max {np.abs(p - ref_p).max()}"
    assert np.array_equal(v.view(np.uint32), ref_v.view(np.uint32)), f"velocities differ: max {np.abs(v - ref_v).max()}"
    assert np.array_equal(d.view(np.uint32), ref_d.view(np.uint32))


def test_grid_layout_geometry():
    from tiles_reference import in_rect, rects_touch

    pos, _ = dam_break(2.0)
    lay = GridLayout.quantile(pos, 2, 2)
    rects = lay.rects()
    assert len(rects) == 4 and rects[0][0] == 0 and rects[3][1] == 65536 and rects[0][2] == 0 and rects[1][3] == 65536
    cx, cy = cell_coord(pos, 0), cell_coord(pos, 1)
    owner = sum(in_rect(cx, cy, r).astype(int) for r in rects)
    assert (owner == 1).all(), "the rectangles partition the domain"
    counts = [int(in_rect(cx, cy, r).sum()) for r in rects]
    assert max(counts) < 1.1 * min(counts), "quantile cuts balance the tiles"
    for a in range(4):
        assert sum(rects_touch(rects[a], rects[b], 8) for b in range(4) if b != a) == 3  # edge + corner neighbours


def test_tiles_2x2_match_single_domain():
    """SURVEY.md 8(e) "4 GPUs: 2x2 tiles": columns cut again across, every tile has two edge neighbours and a corner neighbour;
    particles migrate across both cuts and diagonally.  Same bar as the strips."""
    from tile_oracle_backend import OracleTileBackend

    pos, boundary = dam_break(2.0)
    steps = 60
    ref, rstats, _ = single_domain(pos, boundary, steps)
    outs, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 4, None, steps, halo=8, layout=lambda: GridLayout.quantile(pos, 2, 2))
    for s in range(steps):
        for r in range(4):
            st = outs[r][1][s]
            assert st["density_iterations"] == rstats[s]["density_iterations"] and st["divergence_iterations"] == rstats[s]["divergence_iterations"]
            assert np.float32(st["dt"]) == np.float32(rstats[s]["dt"]) and np.float32(st["vmax"]) == np.float32(rstats[s]["vmax"])
    p, v, d = merge_owned(outs, len(pos))
    np.testing.assert_allclose(p, ref["pos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v, ref["vel"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(d, ref["density"], rtol=1e-5)


def test_tiles_2x2_impact_migration_rebalance():
    """Fixed 3+2 iterations (warm starts, budget-triggered extra exchanges) into the impact with the cuts re-partitioned every
    4 steps, 2x2 tiles vs ONE tile of the same code path.  (The order inside a cell differs between tilings once particles have
    crossed an x-cut; the impact amplifies that round-off chaotically — by step 150 to 1e-5 in position for x-strips and 2x2 tiles
    alike — so the comparison stops at step 100.)"""
    from tile_oracle_backend import OracleTileBackend

    pos, boundary = dam_break(2.0)
    steps = 100
    one, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 1, 1, steps, fixed=(3, 2))
    four, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 4, None, steps, fixed=(3, 2), halo=10, rebalance_every=4,
                                 layout=lambda: GridLayout.quantile(pos, 2, 2))
    states = run_tiles_threaded.final_cuts
    assert all(st == states[0] for st in states) and states[0][1] > 3, "all ranks hold the same, re-partitioned layout"
    p1, v1, _ = merge_owned(one, len(pos))
    p4, v4, _ = merge_owned(four, len(pos))
    np.testing.assert_allclose(p4, p1, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(v4, v1, rtol=1e-4, atol=1e-3)
    lay = GridLayout.quantile(pos, 2, 2)
    own0 = np.array([[in_rect_one(pos[i], r) for r in lay.rects()].index(True) for i in range(0, len(pos), 7)])
    lay2 = GridLayout(states[0][0][0], states[0][0][1:])
    own1 = np.array([[in_rect_one(p4[i], r) for r in lay2.rects()].index(True) for i in range(0, len(pos), 7)])
    assert (own0 != own1).sum() > 0, "particles changed owner"


def in_rect_one(p, rect):
    from tiles_reference import in_rect

    q = np.asarray(p, np.float32).reshape(1, 2)
    return bool(in_rect(cell_coord(q, 0), cell_coord(q, 1), rect)[0])


GLOO_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["REPO_ROOT"], "tests"))
import torch, torch.distributed as dist
import yasph2d_amd as y
from tiles_reference import ShmComm, TiledDFSPH, TorchComm, cell_coord, quantile_cuts
from tile_oracle_backend import OracleTileBackend
from util import dam_break
dist.init_process_group("gloo")
pos, boundary = dam_break(1.0)
cuts = quantile_cuts(cell_coord(pos, 1), dist.get_world_size())
comm = (ShmComm(dist, torch.device("cpu"), "t" + os.environ["MASTER_PORT"]) if os.environ["COMM"] == "shm"
        else TorchComm(dist, torch.device("cpu")))
t = TiledDFSPH(OracleTileBackend(), comm, 1, cuts, halo=16)
t.setup(pos, None, None, boundary)
timer = y.TimeManager()
stats = [t.step(timer) for _ in range(int(os.environ["STEPS"]))]
d = t.download_owned()
np.savez(os.path.join(os.environ["OUT_DIR"], f"rank{dist.get_rank()}.npz"), pos=d["pos"], vel=d["vel"], ids=d["ids"], density=d["density"],
         dt_ns=np.array([s["dt_ns"] for s in stats]), Id=np.array([s["density_iterations"] for s in stats]))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("comm", ["torch", "shm"])
def test_tiles_gloo_world_size_2(tmp_path, comm):
    """The N > 1 launch path of bench.py: one process per rank, torch.distributed (gloo here, RCCL on the GPUs); scalars over
    torch.distributed or over the shared-memory all-reduce (bench.py's default)."""
    import subprocess

    steps = 40
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    env = dict(os.environ, REPO_ROOT=os.path.dirname(HERE), OUT_DIR=str(tmp_path), STEPS=str(steps), OMP_NUM_THREADS="1", COMM=comm)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                           "--master-port", "29531", str(script)], env=env, timeout=600)
    pos, boundary = dam_break(1.0)
    ref, rstats, timer_ns = single_domain(pos, boundary, steps)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    ids = np.concatenate([p["ids"] for p in parts])
    assert sorted(ids.tolist()) == list(range(len(pos)))
    got = np.zeros_like(ref["pos"])
    got[ids] = np.concatenate([p["pos"] for p in parts])
    np.testing.assert_allclose(got, ref["pos"], rtol=1e-5, atol=1e-6)
    for p in parts:
        assert int(p["dt_ns"][-1]) == timer_ns
        np.testing.assert_array_equal(p["Id"], [s["density_iterations"] for s in rstats])


def _shm_worker(rank, world, name, q):
    import ctypes as C

    from yasph2d_amd import _lib

    L = _lib.lib()
    h = L.sphx_shm_open(name.encode(), rank, world)
    assert h
    buf_in, buf_out = (C.c_double * 8)(), (C.c_double * 8)()
    res = []
    for it in range(2000):
        buf_in[0] = float(rank * 1000 + it)
        buf_in[1] = 0.1 * (rank + 1) + it
        assert L.sphx_shm_allreduce(h, buf_in, 2, it & 1, buf_out) == 0
        res.append((buf_out[0], buf_out[1]))
    # the all-gather the halo exchange publishes its per-peer record counts with (round 6), interleaved with all-reduces
    gat = (C.c_double * (8 * world))()
    rows = []
    for it in range(500):
        for k in range(8):
            buf_in[k] = float(rank * 100 + k * 7 + it)
        assert L.sphx_shm_allgather(h, buf_in, 8, gat) == 0
        rows.append([gat[j] for j in range(8 * world)])
        if it % 3 == 0:
            assert L.sphx_shm_allreduce(h, buf_in, 1, 0, buf_out) == 0
    L.sphx_shm_close(h)
    q.put((rank, (res, rows)))


def test_shm_allreduce_four_processes():
    """sphx_shm_allreduce: every rank gets identical bits, combined in rank order, for 2000 back-to-back rounds of alternating ops;
    sphx_shm_allgather: every rank gets every rank's row, 500 rounds interleaved with all-reduces."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    world, name = 4, f"pytest{os.getpid()}"
    q = ctx.Queue()
    ps = [ctx.Process(target=_shm_worker, args=(r, world, name, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for it in range(2000):
        a = [float(r * 1000 + it) for r in range(world)]
        b = [0.1 * (r + 1) + it for r in range(world)]
        if it & 1:
            want = (max(a), max(b))
        else:
            sa, sb = a[0], b[0]
            for r in range(1, world):
                sa, sb = sa + a[r], sb + b[r]
            want = (sa, sb)
        for r in range(world):
            assert got[r][0][it] == want
    for it in range(500):
        want_rows = [float(r * 100 + k * 7 + it) for r in range(world) for k in range(8)]
        for r in range(world):
            assert got[r][1][it] == want_rows


def _shm_victim_worker(rank, world, name, q, die_at, mode):
    """Ranks run all-reduce rounds; rank 1 stops arriving at round `die_at` (mode "kill": the process dies without a word;
    "abort": it reports its failure with sphx_shm_abort; "close": it closes the segment and leaves)."""
    import ctypes as C
    import time

    os.environ["SPHX_SHM_TIMEOUT_S"] = "2"
    from yasph2d_amd import _lib

    L = _lib.lib()
    h = L.sphx_shm_open(name.encode(), rank, world)
    assert h
    buf_in, buf_out = (C.c_double * 8)(), (C.c_double * 8)()
    for it in range(1000):
        if rank == 1 and it == die_at:
            if mode == "kill":
                os._exit(17)
            if mode == "abort":
                L.sphx_shm_abort(h)
                q.put((rank, "aborted", it, 0.0))
                return
            L.sphx_shm_close(h)
            q.put((rank, "closed", it, 0.0))
            return
        buf_in[0] = float(it)
        t0 = time.perf_counter()
        rc = L.sphx_shm_allreduce(h, buf_in, 1, 0, buf_out)
        if rc:
            q.put((rank, "error %d" % rc, it, time.perf_counter() - t0))
            L.sphx_shm_close(h)
            return
        assert buf_out[0] == float(it) * world
    q.put((rank, "finished", 1000, 0.0))


@pytest.mark.parametrize("mode", ["kill", "abort", "close"])
def test_shm_allreduce_survives_a_lost_rank(mode):
    """One of four ranks stops arriving mid-run: the others get SPHX_ERR_NOT_READY from the very round it missed — after the
    time-out when it was killed, at once when it said so — instead of hanging."""
    import multiprocessing as mp

    from yasph2d_amd import _lib

    ctx = mp.get_context("spawn")
    world, name, die_at = 4, f"pytestlost{os.getpid()}{mode}", 137
    q = ctx.Queue()
    ps = [ctx.Process(target=_shm_victim_worker, args=(r, world, name, q, die_at, mode)) for r in range(world)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(world - (1 if mode == "kill" else 0)):
        r, what, it, waited = q.get(timeout=60)
        got[r] = (what, it, waited)
    for p in ps:
        p.join(30)
    assert ps[1].exitcode == (17 if mode == "kill" else 0)
    for r in (0, 2, 3):
        what, it, waited = got[r]
        assert what == "error %d" % _lib.ERR_NOT_READY and it == die_at, got
        assert waited < (4.0 if mode == "kill" else 1.0), got  # time-out 2 s / released at once
        assert ps[r].exitcode == 0


def _shm_stale_worker(rank, world, name, q, delay):
    import ctypes as C
    import time

    os.environ["SPHX_SHM_TIMEOUT_S"] = "20"
    from yasph2d_amd import _lib

    time.sleep(delay)
    L = _lib.lib()
    h = L.sphx_shm_open(name.encode(), rank, world)
    assert h
    a, b = (C.c_double * 8)(), (C.c_double * 8)()
    a[0] = rank + 1.0
    rc = L.sphx_shm_allreduce(h, a, 1, 0, b)
    L.sphx_shm_close(h)
    q.put((rank, rc, b[0]))


def test_shm_open_ignores_a_stale_segment():
    """A segment a crashed run left under the same name (valid magic, old arrivals) must not capture the ranks that get there before
    rank 0 has replaced it (RcclComm::open could put ranks on different segments)."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    name = f"pyteststale{os.getpid()}"
    # a complete earlier "run" whose rank 0 never unlinked: world 1, then the file is left behind on purpose
    import ctypes as C

    from yasph2d_amd import _lib

    L = _lib.lib()
    h = L.sphx_shm_open(name.encode(), 0, 1)
    assert h
    a, b = (C.c_double * 8)(), (C.c_double * 8)()
    for _ in range(5):
        assert L.sphx_shm_allreduce(h, a, 1, 0, b) == 0
    # (no sphx_shm_close: the segment stays in /dev/shm like after a crash)
    assert os.path.exists("/dev/shm/sphx_" + name)
    try:
        world = 3
        q = ctx.Queue()
        # ranks 1 and 2 arrive first and find the stale segment; rank 0 comes a second later
        ps = [ctx.Process(target=_shm_stale_worker, args=(r, world, name, q, 1.0 if r == 0 else 0.0)) for r in range(world)]
        for p in ps:
            p.start()
        got = dict((r, (rc, v)) for r, rc, v in (q.get(timeout=60) for _ in range(world)))
        for p in ps:
            p.join(30)
            assert p.exitcode == 0
        assert all(got[r] == (0, 6.0) for r in range(world)), got
    finally:
        if os.path.exists("/dev/shm/sphx_" + name):
            os.unlink("/dev/shm/sphx_" + name)


def _rccl_bringup_worker(rank, world, job, q, env):
    import ctypes as C
    import time

    os.environ.update(env)
    os.environ["SPHX_SHM_TIMEOUT_S"] = "20"
    import yasph2d_amd as y
    from yasph2d_amd import _lib

    L = _lib.lib()
    h = C.c_void_p()
    t0 = time.perf_counter()
    rc = L.sphx_multi_create_rank(C.byref(y.default_params()), 0, None, job.encode(), rank, world, None, C.byref(h))
    q.put((rank, rc, L.sphx_multi_last_error(None).decode(), time.perf_counter() - t0))


def test_builtin_transport_fails_on_every_rank_when_one_rank_cannot_load_rccl():
    """A rank that cannot bring RCCL up must not leave the others inside ncclCommInitRank or the shared-memory barrier: every
    rank returns an error, promptly, with a message that says what happened.  Runs without a GPU: the
    failure is injected before anything touches the device."""
    import multiprocessing as mp

    from yasph2d_amd import _lib

    ctx = mp.get_context("spawn")
    world, job = 3, f"pytestrccl{os.getpid()}"
    q = ctx.Queue()
    ps = [ctx.Process(target=_rccl_bringup_worker, args=(r, world, job, q, {"SPHX_TEST_FAIL_RCCL_LOAD": "1"})) for r in range(world)]
    for p in ps:
        p.start()
    got = {r: (rc, msg, t) for r, rc, msg, t in (q.get(timeout=120) for _ in range(world))}
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert all(got[r][0] == _lib.ERR_HIP for r in range(world)), got
    assert "injected" in got[1][1], got
    for r in (0, 2):
        assert "other rank(s) failed" in got[r][1] and "librccl" in got[r][1], got
    assert max(t for _, _, t in got.values()) < 15.0, got


def test_rebalance_cuts_invariants_random():
    """Whatever the loads: cuts stay strictly increasing with the outer ones fixed, no cut moves further than max_shift, interior
    tiles keep two halo widths (+2), and the function is a pure function of its inputs."""
    from tiles_reference import rebalance_cuts

    rng = np.random.default_rng(11)
    for _ in range(2000):
        W = int(rng.integers(2, 9))
        halo = int(rng.integers(2, 17))
        widths = rng.integers(2 * halo + 2, 400, W - 2) if W > 2 else np.zeros(0, int)
        first = int(rng.integers(1000, 30000))
        cuts = [0, first] + [int(first + w) for w in np.cumsum(widths)] + [65536]
        counts = [float(c) for c in rng.integers(0, 100000, W)]
        max_shift = int(rng.integers(1, halo + 1))
        new = rebalance_cuts(cuts, counts, halo, float(rng.uniform(1, 500)), max_shift)
        assert new[0] == 0 and new[-1] == 65536 and len(new) == len(cuts)
        assert all(b > a for a, b in zip(new, new[1:]))
        assert all(abs(a - b) <= max_shift for a, b in zip(new, cuts))
        assert all(new[r + 1] - new[r] >= 2 * halo + 2 for r in range(1, W - 1))


def test_grid_layout_partitions_random_clouds():
    from tiles_reference import in_rect

    rng = np.random.default_rng(12)
    for _ in range(20):
        n = int(rng.integers(2000, 20000))
        pos = (rng.normal(0, 1, (n, 2)) * rng.uniform(0.2, 3.0, 2) + rng.uniform(-5, 5, 2)).astype(np.float32)
        nx, ny = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        lay = GridLayout.quantile(pos, nx, ny)
        rects = lay.rects()
        cx, cy = cell_coord(pos, 0), cell_coord(pos, 1)
        owner = sum(in_rect(cx, cy, r).astype(int) for r in rects)
        assert (owner == 1).all()
        # the rectangles tile the whole domain
        area = sum((r[1] - r[0]) * (r[3] - r[2]) for r in rects)
        assert area == 65536 * 65536
