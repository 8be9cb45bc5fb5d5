"""Randomised parity: irregular particle clouds (clusters dense enough to hit the 64-neighbour cap, separated blobs that leave
empty 64x64-cell blocks in the directory, fluid next to cell 0 of the Morton domain, random boundary pieces, random velocities)
stepped with fixed iterations on the GPU and in the oracle.  Everything bit-identical, lists included."""
import numpy as np
import pytest
from util import assert_bits_equal, assert_same_neighbors

import yasph2d_amd as y
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def scene(seed, dense=False):
    rng = np.random.default_rng(seed)
    parts, vels, boundary = [], [], []
    slots = rng.permutation(9)[: int(rng.integers(1, 4))]  # blobs on distinct slots of a 3x3 raster 3 m apart (several 64-cell blocks)
    for sl in slots:
        c = (np.array([sl % 3, sl // 3], np.float32) * np.float32(3.0) + rng.uniform(-0.5, 0.5, 2).astype(np.float32)).astype(np.float32)
        n = int(rng.integers(200, 1500))
        spacing = np.float32(rng.choice([0.004, 0.008, 0.0111, 0.02]) if dense else rng.choice([0.0111, 0.013, 0.02]))  # 0.004: neighbour cap
        side = int(np.ceil(np.sqrt(n)))
        g = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:n].astype(np.float32)
        p = c + g * spacing + rng.uniform(0, 0.3 if dense else 0.1, (n, 2)).astype(np.float32) * spacing
        parts.append(p.astype(np.float32))
        vels.append((rng.normal(0, 0.05, (n, 2)) + rng.normal(0, 1.0, 2)).astype(np.float32))
        if rng.random() < 0.7:  # a floor segment one particle spacing below the blob (dense: right through it, for the static cap)
            w = side * spacing
            x0, x1 = c[0] + rng.uniform(-0.2, 0.4) * w, c[0] + rng.uniform(0.6, 1.2) * w
            yb = c[1] + (np.float32(0.3) * w if dense else -np.float32(1.0) * np.float32(0.0111))
            xs = np.arange(x0, x1, 0.01, dtype=np.float32)
            for row in range(int(rng.integers(1, 4))):
                boundary.append(np.stack([xs, np.full_like(xs, yb - np.float32(0.01) * row)], -1))
    if seed % 3 == 0:  # a sheet hugging the lower-left corner of the Morton domain (cells 0, 1, ...): the reference's rim behaviour
        n = 300
        side = 18
        g = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:n].astype(np.float32)
        parts.append((np.float32(-100.0) + np.float32(0.002) + g * np.float32(0.0111)).astype(np.float32))
        vels.append(np.abs(rng.normal(0, 0.3, (n, 2))).astype(np.float32))
    pos, vel = np.concatenate(parts), np.concatenate(vels)
    boundary = np.concatenate(boundary).astype(np.float32) if boundary else np.zeros((0, 2), np.float32)
    return pos, vel, boundary


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_dense_clusters_neighbor_lists(seed):
    """Clusters at 0.4 x the usual spacing: ~80 candidates inside h, the 64-neighbour cap and the would-be panic of
    neighborhood_search.rs:373 are exercised; lists, counts and flags equal the oracle's (no stepping: such a cloud explodes)."""
    pos, vel, boundary = scene(100 + seed, dense=True)
    for span in (0, 64):
        p = y.default_params()
        p.list_span_limit = span
        ctx = y.SphxContext(p)
        o = Oracle()
        if len(boundary):
            ctx.set_boundary(boundary)
            o.set_boundary(boundary)
        ctx.upload(pos, vel)
        o.set_particles(pos, vel)
        o.update_neighborhood()
        try:
            ctx.update_neighborhood()
        except y.SphxError as e:
            assert e.code == y._lib.ERR_NEIGHBOR_PANIC and (o.neighbor_flags() & 2)
            continue
        assert not (o.neighbor_flags() & 2)
        c, _, _ = ctx.download_neighbors()
        assert_same_neighbors(ctx.download_neighbors(), o.neighbors())
        ctx.update_densities()
        o.update_densities()
        assert_bits_equal(ctx.download()["density"], o.densities(), "densities")


# seed 3 is left out here: that cloud blows up within a few steps (velocities of 1e5 m/s); once a particle outruns the cell directory
# it is kept without neighbours for a build (SPHX_FLAG_STRAY_PARTICLES), which is no longer the reference's arithmetic — see
# test_exploding_cloud_is_survived
@pytest.mark.parametrize("seed", [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15])
@pytest.mark.parametrize("span", [0, 64])
def test_random_scene(seed, span):
    pos, vel, boundary = scene(seed)
    p = y.default_params(fixed_iterations=(2, 2))
    p.list_span_limit = span  # 64: many waves on the 32-bit list fallback
    ctx = y.SphxContext(p)
    o = Oracle()
    o.set_fixed_iterations(2, 2)
    if len(boundary):
        ctx.set_boundary(boundary)
        o.set_boundary(boundary)
    ctx.upload(pos, vel)
    o.set_particles(pos, vel)
    o.update_neighborhood()
    try:
        ctx.update_neighborhood()
    except y.SphxError as e:
        # 64 dynamic neighbours and a static hit: the reference panics at neighborhood_search.rs:373; the oracle notes it in bit 1
        assert e.code == y._lib.ERR_NEIGHBOR_PANIC and (o.neighbor_flags() & 2)
        return
    assert not (o.neighbor_flags() & 2)
    assert_same_neighbors(ctx.download_neighbors(), o.neighbors())
    timer = y.TimeManager(fixed_ns=50_000)  # a fixed small step: random clouds are far from equilibrium
    o.timer_fixed(50_000)
    diam = np.float32(0.01)
    capped = False
    for s in range(12):
        try:
            vmax = ctx.step_begin(timer.simulation_step(), timer.law(diam))
            st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(diam, vmax)))
        except y.SphxError as e:
            assert e.code == y._lib.ERR_NEIGHBOR_PANIC
            o.dfsph_step()
            assert o.neighbor_flags() & 2  # the oracle saw the same would-be panic in this step
            return
        so = o.dfsph_step()
        assert not (o.neighbor_flags() & 2)
        capped |= bool(st["flags"] & y.FLAG_NEIGHBOR_CAP)
        assert st["neighbor_entries"] == so["neighbor_entries"], (seed, s)
        assert np.float32(vmax) == np.float32(so["vmax"]), (seed, s)
    d = ctx.download()
    np.testing.assert_array_equal(d["ids"], o.ids())
    assert_bits_equal(d["pos"], o.positions(), f"seed {seed} positions")
    assert_bits_equal(d["vel"], o.velocities(), f"seed {seed} velocities")
    assert_bits_equal(d["density"], o.densities(), f"seed {seed} densities")
    ss = ctx.download_solver_state()
    assert_bits_equal(ss["kappa"], o.kappa(), "kappa")
    assert_bits_equal(ss["stiffness"], o.stiffness(), "stiffness")
    assert_same_neighbors(ctx.download_neighbors(), o.neighbors())


@pytest.mark.parametrize("seed", [0, 1, 2, 4, 5, 6, 7, 8])
def test_random_scene_wcsph(seed):
    """The same clouds through the WCSPH step (adaptive timer, cfl 0.2): leap frog, Poly6 densities, Tait/Spiky/XSPH/boundary forces."""
    pos, vel, boundary = scene(seed)
    ctx = y.SphxContext()
    o = Oracle()
    timer = y.TimeManager(cfl_factor=0.2)
    o.timer_adaptive(timer.timestep_max_ns, timer.timestep_min_ns, 0.2)
    if len(boundary):
        ctx.set_boundary(boundary)
        o.set_boundary(boundary)
    ctx.upload(pos, vel)
    o.set_particles(pos, vel)
    diam = np.float32(0.01)
    for s in range(25):
        vmax = ctx.wcsph_step_begin(timer.simulation_step())
        dt_ns = timer.update_simulation_step(diam, vmax)
        ctx.wcsph_step_finish(y.duration_as_secs_f32(dt_ns))
        so = o.wcsph_step()
        assert dt_ns == o.timer_step_ns(), (seed, s)
        assert np.float32(vmax) == np.float32(so["vmax"]), (seed, s)
    d = ctx.download()
    np.testing.assert_array_equal(d["ids"], o.ids())
    assert_bits_equal(d["pos"], o.positions(), f"seed {seed} positions")
    assert_bits_equal(d["vel"], o.velocities(), f"seed {seed} velocities")
    assert_bits_equal(d["density"], o.densities(), f"seed {seed} densities")


def test_exploding_cloud_is_survived():
    """Seed 3 blows up (velocities of 1e5 m/s within a few steps).  No parity claim there — the device keeps every particle, reports
    SPHX_FLAG_STRAY_PARTICLES when one outruns the cell directory, and carries on like the reference does."""
    pos, vel, boundary = scene(3)
    ctx = y.SphxContext(y.default_params(fixed_iterations=(2, 2)))
    ctx.set_boundary(boundary)
    ctx.upload(pos, vel)
    timer = y.TimeManager(fixed_ns=50_000)
    diam = np.float32(0.01)
    flags = 0
    for _ in range(12):
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(diam))
        st = ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(diam, vmax)))
        flags |= st["flags"]
    d = ctx.download()
    assert sorted(d["ids"].tolist()) == list(range(len(pos)))
    assert np.isfinite(d["pos"]).all()


@pytest.mark.parametrize("seed", [1, 2, 5, 8, 10, 13])
def test_random_scene_long_adaptive(seed):
    """The same clouds with the adaptive timer and adaptive iteration counts for 400 steps: the blobs fall onto their floor segments
    (or into the void), warm starts and multi-iteration loops occur, the cell directory follows the motion."""
    pos, vel, boundary = scene(seed)
    ctx, o = y.SphxContext(), Oracle()
    if len(boundary):
        ctx.set_boundary(boundary)
        o.set_boundary(boundary)
    ctx.upload(pos, vel)
    o.set_particles(pos, vel)
    timer = y.TimeManager()
    diam = np.float32(0.01)
    for s in range(400):
        vmax = ctx.step_begin(timer.simulation_step(), timer.law(diam))
        dt_ns = timer.update_simulation_step(diam, vmax)
        st = ctx.step_finish(y.duration_as_secs_f32(dt_ns))
        so = o.dfsph_step()
        assert dt_ns == o.timer_step_ns(), (seed, s)
        for k in ("density_iterations", "divergence_iterations", "warmstart_density", "warmstart_divergence", "neighbor_entries"):
            assert st[k] == so[k], (seed, s, k, st[k], so[k])
    d = ctx.download()
    np.testing.assert_array_equal(d["ids"], o.ids())
    assert_bits_equal(d["pos"], o.positions(), f"seed {seed} positions")
    assert_bits_equal(d["vel"], o.velocities(), f"seed {seed} velocities")
    ss = ctx.download_solver_state()
    assert_bits_equal(ss["kappa"], o.kappa(), "kappa")
    assert_bits_equal(ss["stiffness"], o.stiffness(), "stiffness")
