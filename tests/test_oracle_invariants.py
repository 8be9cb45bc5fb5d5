"""Oracle-INDEPENDENT checks of oracle/sph_oracle.cpp.

The DFSPH half of the oracle cannot be pinned to the reference here (no Rust toolchain, the reference holds no solver fixtures:
DESIGN.md §5, "parity unpinned"), and the HIP kernels are compared against the oracle only — a transcription slip shared by both
would be invisible to every parity test.  These tests hold the oracle to properties that follow from the reference's formulas
themselves, not from any restatement of them:

  * the pressure corrections exchange momentum pairwise: (k_i + k_j) grad W_ij is antisymmetric in (i, j) (dfsph.rs:151, :184,
    :305, :335), so constant-density iterations, divergence iterations and both warm starts leave sum(v) of a boundary-free blob
    unchanged up to round-off;
  * a boundary-free blob at rest density in free fall gains exactly g*dt of velocity per particle and step (XSPH is antisymmetric
    when all densities are equal, xsph.rs:21-23; the pressure solve is idle);
  * the residual of a solver loop falls from iteration to iteration (dfsph.rs:217-226, :372-381);
  * a mirror-symmetric scene stays mirror-symmetric, and a scene with x and y exchanged (gravity turned with it) evolves into the
    exchanged result — until the first warm start: the reference's slot-bound warm-start arrays (dfsph.rs:512)
    tie the physics to the Morton order, which is not mirror-symmetric, so from there on only the ordering-independent part holds;
  * the scheme only dissipates: total mechanical energy of the reference scene never rises above its initial value, and the
    fluid stays inside its container.

A sign or ordering slip in a gradient, a swapped component, a missing mass factor or a wrong dt in a correction breaks at least
one of them by orders of magnitude more than the tolerances below.
"""
import numpy as np
from tile_oracle_backend import OracleTileBackend
from util import dam_break

import yasph2d_amd as y
from oracle.oracle import Oracle

EPS = float(np.finfo(np.float32).eps)


def blob(ns=30, spacing=0.0111, center=(1.0, 1.0), seed=1):
    rng = np.random.default_rng(seed)
    g = (np.arange(ns) - 0.5 * (ns - 1)) * spacing
    x, yy = np.meshgrid(g, g)
    p = np.stack([x.ravel(), yy.ravel()], 1) + rng.uniform(-0.05, 0.05, (ns * ns, 2)) * spacing
    return (p + np.array(center)).astype(np.float32)


def whole_domain_backend(pos, vel):
    b = OracleTileBackend()
    b.configure((0, 65536, 0, 65536), 8, [])
    b.upload(pos, vel, np.arange(len(pos)))
    b.regrid()
    return b


def vsum(b):
    v = b.o.velocities().astype(np.float64)
    return v.sum(0), np.abs(v).sum()


def test_density_iterations_and_warm_start_conserve_momentum():
    pos = blob()
    vel = (-(pos - pos.mean(0)) * 3.0).astype(np.float32)  # converging field: the blob is being compressed
    b = whole_domain_backend(pos, vel)
    dt = np.float32(1.0 / 360.0)
    s0, a0 = vsum(b)
    tol = 8 * EPS * a0  # round-off of ~N additions of O(|v|) terms; a sign slip moves sum(v) by O(sum |dv|) = 1e7 times more
    prev = b.o.velocities().copy()
    residuals = []
    for it in range(8):
        s, n = b.iteration(False, dt, it == 0)
        residuals.append(s / n)
        v = b.o.velocities()
        moved = np.abs(v.astype(np.float64) - prev).sum()
        assert moved > 1e4 * tol, "the iteration must actually change the velocities for this test to mean anything"
        assert np.abs(vsum(b)[0] - s0).max() < tol, f"iteration {it}: sum(v) moved by {vsum(b)[0] - s0}"
        prev = v.copy()
    assert all(b2 < a2 for a2, b2 in zip(residuals, residuals[1:])), f"residual must fall monotonically: {residuals}"
    assert np.abs(b.o.kappa()).max() > 0
    b.warmstart(False, dt)  # dfsph.rs:163-193 with the accumulated kappa
    assert np.abs(b.o.velocities().astype(np.float64) - prev).sum() > 1e4 * tol
    assert np.abs(vsum(b)[0] - s0).max() < tol


def test_divergence_iterations_and_warm_start_conserve_momentum():
    pos = blob(spacing=0.0085)  # dense enough that (nearly) every particle has >= 9 neighbours (dfsph.rs:261 gate)
    vel = (-(pos - pos.mean(0)) * 3.0).astype(np.float32)
    b = whole_domain_backend(pos, vel)
    assert (b.o.neighbors()[0][:, 1] >= 9).mean() > 0.98
    dt = np.float32(1.0 / 360.0)
    s0, a0 = vsum(b)
    tol = 8 * EPS * a0
    prev = b.o.velocities().copy()
    residuals = []
    for it in range(6):
        s, n = b.iteration(True, dt, it == 0)
        residuals.append(s / n)
        v = b.o.velocities()
        assert np.abs(v.astype(np.float64) - prev).sum() > 1e4 * tol
        assert np.abs(vsum(b)[0] - s0).max() < tol, f"iteration {it}: sum(v) moved by {vsum(b)[0] - s0}"
        prev = v.copy()
    assert all(b2 < a2 for a2, b2 in zip(residuals, residuals[1:])), f"residual must fall monotonically: {residuals}"
    b.warmstart(True, dt)
    assert np.abs(b.o.velocities().astype(np.float64) - prev).sum() > 1e4 * tol
    assert np.abs(vsum(b)[0] - s0).max() < tol


def test_free_fall_gains_exactly_the_gravity_impulse():
    """No boundary, lattice at (clamped) rest density: XSPH pairs cancel, the solver is idle, every step adds g*dt to sum(v)/N."""
    pos = blob(ns=24)
    rng = np.random.default_rng(5)
    vel = rng.normal(0.0, 0.05, pos.shape).astype(np.float32)  # XSPH has something to exchange
    o = Oracle()
    o.set_boundary(np.zeros((0, 2), np.float32))
    o.set_particles(pos, vel)
    n = len(pos)
    for step in range(10):
        ids0 = np.argsort(o.ids())
        v0 = o.velocities().astype(np.float64)[ids0]
        st = o.dfsph_step()
        v1 = o.velocities().astype(np.float64)[np.argsort(o.ids())]
        assert (o.densities() == np.float32(100.0)).all()
        assert np.abs(v1 - v0).sum() > 1e-3, "XSPH must be active"
        gain = (v1.sum(0) - v0.sum(0)) / n
        want = np.array([0.0, -9.81 * st["dt"]])
        assert np.abs(gain - want).max() < 4 * EPS * max(1.0, np.abs(v1).max()), f"step {step}: {gain} vs {want}"


def _box_boundary():
    w = y.FluidParticleWorld()
    w.add_boundary_thick_line((-0.6, 0.0), (0.6, 0.0), 3)
    w.add_boundary_thick_line((-0.6, 0.0), (-0.6, 1.0), 3)
    w.add_boundary_thick_line((0.6, 0.0), (0.6, 1.0), 3)
    return w.boundary_particles


def _box_fluid():
    w = y.FluidParticleWorld()
    w.add_fluid_rect(-0.25, 0.05, 0.5, 0.5, 0.05)
    return w.positions


def _by_id(o):
    inv = np.argsort(o.ids())
    return o.positions()[inv], o.velocities()[inv]


def test_mirror_symmetric_scene_stays_mirror_symmetric():
    def symm(p):
        h = p[p[:, 0] > 0]
        return np.concatenate([h, h * np.array([-1, 1], np.float32)]).astype(np.float32), len(h)

    pos, nh = symm(_box_fluid())
    bnd, _ = symm(_box_boundary())
    o = Oracle()
    o.set_boundary(bnd)
    o.set_particles(pos)
    solver_worked = False
    steps = 0
    for _ in range(80):
        st = o.dfsph_step()
        if st["warmstart_density"] or st["warmstart_divergence"]:
            break  # slot-bound warm start: tied to the (asymmetric) Morton order from here on
        steps += 1
        solver_worked |= st["divergence_iterations"] > 1 or st["density_iterations"] > 1
        p, v = _by_id(o)
        a, b, va, vb = p[:nh], p[nh:], v[:nh], v[nh:]
        assert max(np.abs(a[:, 0] + b[:, 0]).max(), np.abs(a[:, 1] - b[:, 1]).max()) < 1e-6
        assert max(np.abs(va[:, 0] + vb[:, 0]).max(), np.abs(va[:, 1] - vb[:, 1]).max()) < 1e-5
    assert steps > 30 and solver_worked, "the block must have reached the floor (pressure solve active) while still symmetric"


def test_exchanging_x_and_y_exchanges_the_result():
    pos, bnd = _box_fluid(), _box_boundary()
    a = Oracle()
    a.set_boundary(bnd)
    a.set_particles(pos)
    b = Oracle(gravity=(-9.81, 0.0))
    b.set_boundary(bnd[:, ::-1].copy())
    b.set_particles(pos[:, ::-1].copy())
    steps = 0
    for _ in range(80):
        sa, sb = a.dfsph_step(), b.dfsph_step()
        if sa["warmstart_density"] or sa["warmstart_divergence"] or sb["warmstart_density"] or sb["warmstart_divergence"]:
            break
        steps += 1
        assert sa["density_iterations"] == sb["density_iterations"] and sa["divergence_iterations"] == sb["divergence_iterations"]
        assert abs(sa["dt"] - sb["dt"]) <= 2e-6 * sa["dt"]
        pa, va = _by_id(a)
        pb, vb = _by_id(b)
        assert np.abs(pa - pb[:, ::-1]).max() < 1e-6 and np.abs(va - vb[:, ::-1]).max() < 1e-5
    assert steps > 30


def test_energy_only_dissipates_and_the_fluid_stays_in_its_container():
    pos, bnd = dam_break(1.0)
    o = Oracle()
    o.set_boundary(bnd)
    o.set_particles(pos)
    m = float(o.properties()["particle_mass"])

    def energy():
        p, v = o.positions().astype(np.float64), o.velocities().astype(np.float64)
        return m * (0.5 * (v * v).sum(1) + 9.81 * p[:, 1]).sum()

    e0 = energy()
    lo, hi = bnd.min(0), bnd.max(0)
    for step in range(600):
        o.dfsph_step()
        e = energy()
        assert e <= e0 * (1 + 1e-6), f"step {step}: mechanical energy rose to {e / e0} of its initial value"
        if step % 50 == 49:
            p = o.positions()
            assert (p >= lo).all() and (p <= hi).all(), "a particle left the container"
            d = o.densities()
            assert 100.0 <= d.mean() < 103.0
    assert energy() < 0.93 * e0, "viscosity and the impacts must have dissipated energy by now"


def test_all_parallel_variant_of_the_cpu_baseline_computes_the_same_run():
    """bench.py's cpu_baseline reports two variants of the OpenMP restatement: the reference-faithful one and `all_parallel` (the loops
    the reference leaves serial run in parallel too and — round 4 — list space is reserved per thread instead of with one fetch_add per
    particle on a shared counter, the reference's AppendBuffer).  Only the PLACE of a particle's list in the buffer differs (the
    reference's start_index depends on the thread schedule anyway): positions, velocities, densities, iteration counts and the neighbour
    lists themselves are identical, step by step."""
    from oracle.oracle import lib

    L = lib(omp=True)
    pos, boundary = dam_break(1.5)
    runs = []
    for allpar in (0, 1):
        L.orc_set_all_parallel(allpar)
        o = Oracle(omp=True)
        o.set_boundary(boundary)
        o.set_particles(pos)
        stats = [o.dfsph_step() for _ in range(80)]
        c, s, l = o.neighbors()
        runs.append((o.positions().copy(), o.velocities().copy(), o.densities().copy(), [(t["density_iterations"], t["divergence_iterations"]) for t in stats],
                     c.copy(), l.copy(), stats[-1]["neighbor_entries"]))
        del o
    L.orc_set_all_parallel(0)
    a, b = runs
    for k in range(3):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
    assert a[3] == b[3] and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
    assert a[6] == b[6] == int(a[4][:, 1].sum())
