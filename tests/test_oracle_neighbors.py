"""Re-runs the reference's neighbour-search property test (src/sph/neighborhood_search.rs:524-557) on the oracle:
for every particle the dynamic neighbour list equals the brute-force set {i != p : d^2 <= R^2} in ascending index order."""
import numpy as np
import pytest
from util import brute_force_neighbors, uniform_points

from oracle.oracle import Oracle


def _check(pos, radius, sample=None):
    o = Oracle(search_radius=radius)
    o.set_particles(pos)
    o.update_neighborhood()
    p = o.positions()  # sorted in place, like the reference
    counts, start, lists = o.neighbors()
    assert (counts[:, 0] == counts[:, 1]).all()  # no static particles
    idx = range(len(p)) if sample is None else sample
    for i in idx:
        got = lists[int(start[i]):int(start[i + 1])]
        np.testing.assert_array_equal(got, brute_force_neighbors(p, radius, i))
    return o, p, counts


def test_neighbors_contains_neighbors():
    # NUM_POSITIONS = 1000, DENSITY = 10, SEARCH_RADIUS = 1 (neighborhood_search.rs:531-533)
    o, p, counts = _check(uniform_points(1000, 10.0, 123456789), 1.0)
    assert counts[:, 1].mean() > 20  # ~ pi * 10


def test_bench_workload_20000_points():
    # benches/benchmarks/neighborhood_search.rs:10-21
    rng = np.random.default_rng(7)
    o, p, counts = _check(uniform_points(20000, 10.0, 123456789), 1.0, sample=rng.integers(0, 20000, 400))
    first, cidx = o.cells()
    assert first[-1] == 20000 and cidx[-1] == 0xFFFFFFFF  # sentinel, neighborhood_search.rs:161-164
    assert (np.diff(cidx[:-1].astype(np.int64)) > 0).all()  # strictly ascending Morton codes
    assert (np.diff(first.astype(np.int64)) > 0).all()


def test_sort_is_stable_and_a_permutation():
    pos = uniform_points(5000, 10.0, 5)
    o = Oracle(search_radius=1.0)
    o.set_particles(pos)
    o.update_neighborhood()
    ids = o.ids()
    assert sorted(ids.tolist()) == list(range(5000))
    np.testing.assert_array_equal(o.positions(), pos[ids])
    # stable: inside each cell the previous indices ascend
    first, _ = o.cells()
    for a, b in zip(first[:-1], first[1:]):
        assert (np.diff(ids[a:b].astype(np.int64)) > 0).all()


def test_static_neighbors_follow_dynamic_and_cap():
    from util import bench_world

    pos, boundary = bench_world()
    assert len(pos) == 8100  # benches/benchmarks/update_densities.rs:78 (90 x 90)
    o = Oracle()
    o.set_boundary(boundary)
    o.set_particles(pos)
    o.update_neighborhood()
    p, b = o.positions(), o.boundary()
    counts, start, lists = o.neighbors()
    h = o.properties()["smoothing_length"]
    rng = np.random.default_rng(3)
    for i in rng.integers(0, len(p), 300):
        cd, ct = counts[i]
        got = lists[int(start[i]):int(start[i + 1])]
        np.testing.assert_array_equal(got[:cd], brute_force_neighbors(p, h, i))
        d = b - p[i]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
        np.testing.assert_array_equal(got[cd:ct], np.nonzero((d2 <= h * h) & (d2 > np.float32(1e-10)))[0].astype(np.uint32))
    assert counts[:, 1].max() <= 64


@pytest.mark.parametrize("n", [0, 1, 2])
def test_tiny_inputs(n):
    pos = uniform_points(max(n, 1), 10.0, 1)[:n]
    o = Oracle(search_radius=1.0)
    o.set_particles(pos)
    o.update_neighborhood()
    counts, start, lists = o.neighbors()
    assert counts.shape == (n, 2)
    first, cidx = o.cells()
    assert first[-1] == n and cidx[-1] == 0xFFFFFFFF


def test_coincident_points_are_not_neighbors():
    # MIN_DISTANCE filter (neighborhood_search.rs:323,357): d^2 > 1e-10, which also removes self
    pos = np.array([[1.0, 1.0], [1.0, 1.0], [1.5, 1.0], [1.0, 1.000001]], np.float32)
    o = Oracle(search_radius=1.0)
    o.set_particles(pos)
    o.update_neighborhood()
    p = o.positions()
    counts, start, lists = o.neighbors()
    for i in range(4):
        d = p - p[i]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
        expect = np.nonzero((d2 <= 1.0) & (d2 > np.float32(1e-10)))[0]
        np.testing.assert_array_equal(lists[int(start[i]):int(start[i + 1])], expect)


def test_neighbor_cap_64():
    # 100 points inside one radius: every list is truncated at 64 in ascending order and the flag is raised
    rng = np.random.default_rng(11)
    pos = (np.float32(5.0) + rng.random((100, 2), dtype=np.float32) * np.float32(0.2)).astype(np.float32)
    o = Oracle(search_radius=1.0)
    o.set_particles(pos)
    o.update_neighborhood()
    counts, start, lists = o.neighbors()
    assert (counts[:, 1] == 64).all()
    assert o.neighbor_flags() & 1
    p = o.positions()
    for i in (0, 50, 99):
        np.testing.assert_array_equal(lists[int(start[i]):int(start[i + 1])], brute_force_neighbors(p, 1.0, i)[:64])
