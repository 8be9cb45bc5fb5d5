"""Shared helpers for the parity tests: scenes (inputs) and oracle-vs-device comparison."""
import numpy as np


def dam_break(scale=1.0):
    """The reference scene (main.rs:177-196) through the host mirror: (positions, boundary)."""
    import yasph2d_amd as y

    w = y.FluidParticleWorld()
    w.reset_fluid(scale)
    return w.positions, w.boundary_particles


def bench_world():
    """benches/benchmarks/update_densities.rs:72-80: 1x1 m rect with jitter 0.5 + a 20-thick boundary line."""
    import yasph2d_amd as y

    w = y.FluidParticleWorld()
    w.add_fluid_rect(0.0, 0.0, 1.0, 1.0, 0.5)
    w.add_boundary_thick_line((-0.5, 0.5), (1.5, 0.5), 20)
    return w.positions, w.boundary_particles


def uniform_points(n, density, seed):
    """benches/benchmarks/neighborhood_search.rs:10-17 / neighborhood_search.rs:531-538: n uniform points in a
    sqrt(n/density)-sided square (own RNG: the reference's SmallRng stream is not reproducible here)."""
    rng = np.random.default_rng(seed)
    side = np.float32(np.sqrt(np.float32(n) / np.float32(density)))
    return (rng.random((n, 2), dtype=np.float32) * side).astype(np.float32)


def brute_force_neighbors(pos, radius, i):
    d = pos - pos[i]
    d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]  # f32, un-fused like the reference
    m = d2 <= np.float32(radius) * np.float32(radius)
    m[i] = False
    return np.nonzero(m)[0].astype(np.uint32)


def assert_same_neighbors(a, b):
    """a, b = (counts[N,2], start, lists)"""
    ca, sa, la = a
    cb, sb, lb = b
    assert ca.shape == cb.shape
    bad = np.nonzero((ca != cb).any(axis=1))[0]
    assert bad.size == 0, f"neighbor counts differ at {bad[:10]} ({bad.size} particles): {ca[bad[:5]]} vs {cb[bad[:5]]}"
    assert la.shape == lb.shape
    diff = np.nonzero(la != lb)[0]
    assert diff.size == 0, f"neighbor lists differ at {diff.size} entries, first at flat index {diff[:5]}"


def assert_bits_equal(a, b, what):
    """fp32 arrays must be numerically identical element by element (+0 == -0 allowed, NaN never)."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.isfinite(a).all() and np.isfinite(b).all(), f"{what}: non-finite values"
    bad = np.nonzero(a != b)
    if bad[0].size:
        i = tuple(x[0] for x in bad)
        raise AssertionError(f"{what}: {bad[0].size} of {a.size} elements differ; first at {i}: {a[i]!r} vs {b[i]!r}")
