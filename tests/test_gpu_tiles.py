"""Spatial tiles on the GPU: two (three) tiles run as threads of one process on one MI355X (halo records go through device
buffers exactly as they would through RCCL) and must reproduce (a) the oracle-backed tile run of the same tiling BIT-EXACTLY
(same sub-steps, same record order), and (b) the single-context HIP run within fp32 tolerance."""
import numpy as np
import pytest
from test_tiles_cpu import merge_owned, run_tiles_threaded
from util import assert_bits_equal, dam_break

import yasph2d_amd as y
from tiles_reference import GpuTileBackend

pytestmark = pytest.mark.gpu


def gpu_backend(r):
    return GpuTileBackend(y.SphxContext())


@pytest.mark.parametrize("world,axis,halo,fixed,steps", [(2, 1, 16, (0, 0), 80), (3, 1, 8, (0, 0), 40), (2, 0, 6, (0, 0), 40), (2, 1, 10, (3, 2), 60)])
def test_gpu_tiles_bit_exact_vs_oracle_tiles(world, axis, halo, fixed, steps):
    from tile_oracle_backend import OracleTileBackend

    pos, boundary = dam_break(1.0)
    g, cuts = run_tiles_threaded(gpu_backend, pos, boundary, world, axis, steps, halo=halo, fixed=fixed)
    o, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, world, axis, steps, halo=halo, fixed=fixed)
    for r in range(world):
        dg, sg, xg = g[r]
        do, so, xo = o[r]
        assert xg == xo  # same number of halo exchanges (ring budget)
        for a, b in zip(sg, so):
            for k in ("density_iterations", "divergence_iterations", "dt_ns", "n_local", "n_global"):
                assert a[k] == b[k], (r, k, a[k], b[k])
        np.testing.assert_array_equal(dg["ids"], do["ids"])
        assert_bits_equal(dg["pos"], do["pos"], f"rank {r} positions")
        assert_bits_equal(dg["vel"], do["vel"], f"rank {r} velocities")
        assert_bits_equal(dg["density"], do["density"], f"rank {r} densities")
        assert_bits_equal(dg["kappa"], do["kappa"], f"rank {r} kappa")


def test_gpu_tiles_vs_single_context():
    pos, boundary = dam_break(float(np.sqrt(40000 / 4050)))
    steps = 40
    ctx = y.SphxContext()
    ctx.set_boundary(boundary)
    ctx.upload(pos)
    timer = y.TimeManager()
    for _ in range(steps):
        vmax = ctx.step_begin(timer.simulation_step())
        ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(np.float32(0.01), vmax)))
    d = ctx.download()
    inv = np.argsort(d["ids"])
    outs, cuts = run_tiles_threaded(gpu_backend, pos, boundary, 2, 1, steps, halo=16)
    p, v, den = merge_owned(outs, len(pos))
    np.testing.assert_allclose(p, d["pos"][inv], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v, d["vel"][inv], rtol=1e-4, atol=1e-5)
    assert outs[0][1][-1]["dt_ns"] == timer.simulation_step_ns()


def test_gpu_tiles_rebalance_bit_exact_vs_oracle_tiles():
    """Lopsided initial cut + re-partition every 2 steps (cuts move while particles migrate; boundary re-clip margin active):
    the HIP tiles follow the oracle tiles bit for bit, cut for cut."""
    from tile_oracle_backend import OracleTileBackend
    from tiles_reference import cell_coord

    pos, boundary = dam_break(1.0)
    c = cell_coord(pos, 1)
    cuts = [0, int(np.sort(c)[int(0.3 * len(c))]), 65536]
    g, _ = run_tiles_threaded(gpu_backend, pos, boundary, 2, 1, 120, halo=8, cuts=cuts, rebalance_every=2, fixed=(2, 2))
    gc = list(run_tiles_threaded.final_cuts)
    o, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 2, 1, 120, halo=8, cuts=cuts, rebalance_every=2, fixed=(2, 2))
    assert gc == list(run_tiles_threaded.final_cuts) and gc[0][1] > 5
    for r in range(2):
        np.testing.assert_array_equal(g[r][0]["ids"], o[r][0]["ids"])
        for k in ("pos", "vel", "density", "kappa", "stiffness"):
            assert_bits_equal(g[r][0][k], o[r][0][k], f"rank {r} {k}")


def test_gpu_tiles_2x2_bit_exact_vs_oracle_tiles():
    """SURVEY.md 8(e) "4 GPUs: 2x2 tiles": four tile threads on one MI355X (three peers each: two edges and a corner), fixed 3+2
    iterations through the impact (warm starts, diagonal migration, budget-triggered extra exchanges), cuts re-partitioned every
    4 steps: the HIP tiles follow the oracle tiles bit for bit."""
    from test_tiles_cpu import GridLayout
    from tile_oracle_backend import OracleTileBackend

    pos, boundary = dam_break(2.0)
    kw = dict(halo=10, fixed=(3, 2), rebalance_every=4, layout=lambda: GridLayout.quantile(pos, 2, 2), adaptive_halo=True)
    g, _ = run_tiles_threaded(gpu_backend, pos, boundary, 4, None, 150, **kw)
    gl = list(run_tiles_threaded.final_cuts)
    o, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 4, None, 150, **kw)
    assert gl == list(run_tiles_threaded.final_cuts) and gl[0][1] > 3
    merge_owned(g, len(pos))  # every particle owned exactly once
    for r in range(4):
        assert g[r][2] == o[r][2]  # same number of halo exchanges
        np.testing.assert_array_equal(g[r][0]["ids"], o[r][0]["ids"])
        for k in ("pos", "vel", "density", "kappa", "stiffness"):
            assert_bits_equal(g[r][0][k], o[r][0][k], f"rank {r} {k}")


def test_gpu_tiles_adaptive_halo_bit_exact_vs_oracle_tiles():
    """Adaptive ghost band (shrinks to 8 cells while Id = Iv = 1): same exchanges, same bits as the oracle tiles."""
    from tile_oracle_backend import OracleTileBackend

    pos, boundary = dam_break(1.0)
    g, _ = run_tiles_threaded(gpu_backend, pos, boundary, 2, 1, 150, halo=16, adaptive_halo=True, rebalance_every=8)
    hg = [list(h) for h in run_tiles_threaded.halos]
    o, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 2, 1, 150, halo=16, adaptive_halo=True, rebalance_every=8)
    assert hg == [list(h) for h in run_tiles_threaded.halos] and min(hg[0]) == 8
    for r in range(2):
        assert g[r][2] == o[r][2]
        np.testing.assert_array_equal(g[r][0]["ids"], o[r][0]["ids"])
        for k in ("pos", "vel", "density", "kappa", "stiffness"):
            assert_bits_equal(g[r][0][k], o[r][0][k], f"rank {r} {k}")


@pytest.mark.parametrize("seed,nx,ny", [(1, 2, 2), (4, 3, 1), (5, 1, 2), (7, 2, 2)])
def test_gpu_tiles_random_clouds_bit_exact_vs_oracle_tiles(seed, nx, ny):
    """Irregular clouds (separated blobs, floors; tests/test_gpu_random_scenes.py) on quantile grids: tiles of very different
    extent, some nearly empty on one side; adaptive band, re-partitioning.  HIP tiles == oracle tiles, bit for bit."""
    from test_gpu_random_scenes import scene
    from test_tiles_cpu import GridLayout
    from tile_oracle_backend import OracleTileBackend

    pos, vel, boundary = scene(seed)
    if len(boundary) == 0:
        boundary = np.array([[50.0, 50.0]], np.float32)  # the drivers expect a boundary array; one far-away particle
    kw = dict(halo=8, rebalance_every=4, adaptive_halo=True, layout=lambda: GridLayout.quantile(pos, nx, ny))
    g, _ = run_tiles_threaded(gpu_backend, pos, boundary, nx * ny, None, 40, **kw)
    o, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, nx * ny, None, 40, **kw)
    merge_owned(g, len(pos))
    for r in range(nx * ny):
        assert g[r][2] == o[r][2]
        np.testing.assert_array_equal(g[r][0]["ids"], o[r][0]["ids"])
        for k in ("pos", "vel", "density", "kappa", "stiffness"):
            assert_bits_equal(g[r][0][k], o[r][0][k], f"rank {r} {k}")


def test_gpu_tiles_are_deterministic_run_to_run():
    """The same tiled run 25 times: exchanges, iteration counts, local counts and the final bits never vary.  (This is the probe
    that exposed a null-stream memset racing with the next build's histogram — one run in ~30 lost a few ghosts.)"""
    from test_gpu_random_scenes import scene
    from test_tiles_cpu import GridLayout

    pos, _, boundary = scene(7)
    kw = dict(halo=8, rebalance_every=4, adaptive_halo=True, layout=lambda: GridLayout.quantile(pos, 2, 2))

    def summary(out):
        return [(o[2], [(s["density_iterations"], s["divergence_iterations"], s["n_local"], s["n_global"]) for s in o[1]],
                 o[0]["pos"].tobytes(), o[0]["vel"].tobytes()) for o in out]

    ref = None
    for _ in range(25):
        out, _ = run_tiles_threaded(gpu_backend, pos, boundary, 4, None, 40, **kw)
        s = summary(out)
        if ref is None:
            ref = s
        assert s == ref


def test_gpu_tile_that_owns_nothing():
    """A tile whose rectangle holds no fluid at all (all particles on the other side of the cut): it reports 0 owned particles to
    the residual average every iteration — not a stale count — and the other tile reproduces the oracle tiles bit for bit."""
    from tile_oracle_backend import OracleTileBackend
    from tiles_reference import cell_coord

    pos, boundary = dam_break(1.0)
    c = cell_coord(pos, 0)
    cuts = [0, int(c.max()) + 40, 65536]  # everything is left of the cut, 40 cells away from it
    g, _ = run_tiles_threaded(gpu_backend, pos, boundary, 2, 0, 60, halo=8, cuts=cuts)
    o, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 2, 0, 60, halo=8, cuts=cuts)
    assert len(g[1][0]["ids"]) == 0 and len(g[0][0]["ids"]) == len(pos)
    for a, b in zip(g[0][1], o[0][1]):
        for k in ("density_iterations", "divergence_iterations", "dt_ns", "n_global"):
            assert a[k] == b[k]
    assert_bits_equal(g[0][0]["pos"], o[0][0]["pos"], "positions")
    assert_bits_equal(g[0][0]["vel"], o[0][0]["vel"], "velocities")


@pytest.mark.gpu
def test_gpu_empty_tile_receives_its_first_particles():
    """The dam break spreads into a tile that started with nothing: the first records it receives must get cells (its directory is
    empty until then) and stay — owned counts and every particle agree with the oracle tiles bit for bit."""
    from tile_oracle_backend import OracleTileBackend
    from tiles_reference import cell_coord

    pos, boundary = dam_break(1.0)
    c = cell_coord(pos, 0)
    cuts = [0, int(c.max()) + 3, 65536]  # the front crosses the cut after a few dozen steps
    g, _ = run_tiles_threaded(gpu_backend, pos, boundary, 2, 0, 400, halo=8, cuts=cuts)
    o, _ = run_tiles_threaded(lambda r: OracleTileBackend(), pos, boundary, 2, 0, 400, halo=8, cuts=cuts)
    assert len(o[1][0]["ids"]) > 0, "scene does not reach the empty tile: lengthen the run"
    for r in range(2):
        assert np.array_equal(g[r][0]["ids"], o[r][0]["ids"])
        assert_bits_equal(g[r][0]["pos"], o[r][0]["pos"], "positions")
        assert_bits_equal(g[r][0]["vel"], o[r][0]["vel"], "velocities")
    assert len(g[0][0]["ids"]) + len(g[1][0]["ids"]) == len(pos)
