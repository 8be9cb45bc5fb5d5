"""CPU tests of the host-side logic: Duration restatement, TimeManager dt law, scene helpers, and agreement between the
product's host mirror (libsphx.so, no GPU needed for these entry points) and the oracle's independent restatement."""
import itertools
import struct
from fractions import Fraction

import numpy as np
import pytest

import yasph2d_amd as y


def exact_round_ns(f32):
    """Duration::from_secs_f32 semantics (Rust >= 1.63): exact value * 1e9, round half to even."""
    fr = Fraction(float(np.float32(f32))) * 10**9
    q, r = divmod(fr.numerator, fr.denominator)
    twice = 2 * r
    if twice > fr.denominator or (twice == fr.denominator and q % 2 == 1):
        q += 1
    return q


def test_duration_from_secs_f32_exact(oracle_lib, sphx_lib):
    rng = np.random.default_rng(0)
    vals = [0.0, 1.0 / 360.0, 1.0 / 24000.0, 1e-9, 0.5e-9, 1.5e-9, 2.5e-9, 0.006, 1.0, 3.75, 1e-12, 123.456]
    vals += list(10.0 ** rng.uniform(-9, 1, 2000))
    vals += [struct.unpack("f", struct.pack("I", int(b)))[0] for b in rng.integers(0x30000000, 0x42000000, 2000)]
    for v in vals:
        v = np.float32(v)
        want = exact_round_ns(v)
        assert oracle_lib.orc_duration_from_secs_f32(v) == want, v
        assert sphx_lib.sphx_duration_from_secs_f32(v) == want, v


def test_duration_as_secs_f32(oracle_lib, sphx_lib):
    for ns in [0, 1, 41667, 2777778, 999999999, 1000000000, 1500000000, 123456789012]:
        secs, nanos = divmod(ns, 10**9)
        want = np.float32(secs) + np.float32(nanos) / np.float32(1e9)
        assert np.float32(oracle_lib.orc_duration_as_secs_f32(ns)) == want
        assert np.float32(sphx_lib.sphx_duration_as_secs_f32(ns)) == want


def test_app_timer_constants():
    # main.rs:123-124: 1/120/3 s and 1/60/400 s through from_secs_f32
    t = y.TimeManager()
    assert t.timestep_max_ns == exact_round_ns(np.float32(1.0) / np.float32(120.0) / np.float32(3.0)) == 2777778
    assert t.timestep_min_ns == exact_round_ns(np.float32(1.0) / np.float32(60.0) / np.float32(400.0)) == 41667
    assert t.simulation_step_ns() == t.timestep_min_ns  # timemanager.rs:106-109


def test_update_simulation_step_law_matches_oracle(oracle_lib):
    """timemanager.rs:252-279: clamp(cfl*0.4*d/(vmax+1e-5)) to [min, min(max, 2*prev)] — mirror vs oracle vs formula."""
    from oracle.oracle import Oracle

    o = Oracle()
    t = y.TimeManager()
    rng = np.random.default_rng(4)
    prev = t.simulation_step_ns()
    for vmax in list(rng.uniform(0, 20, 300)) + [0.0, 1e-6, 1000.0]:
        vmax = np.float32(vmax)
        d = np.float32(0.01)
        cfl = exact_round_ns(np.float32(1.5) * np.float32(0.4) * d / (vmax + np.float32(0.00001)))
        want = max(t.timestep_min_ns, min(min(t.timestep_max_ns, 2 * prev), cfl))
        got = t.update_simulation_step(d, vmax)
        assert got == want
        assert oracle_lib.orc_timer_update(o.h, d, vmax) == want
        prev = got


def test_target_frame_length_law(oracle_lib):
    """AdaptiveTimeStepTarget::TargetFrameLength (timemanager.rs:268-274), restated literally: the lower bound becomes
    min(timestep_min, total_simulated_time mod target).  Mirror vs oracle vs formula; the law handed to the device carries it."""
    from oracle.oracle import Oracle

    o = Oracle()
    t = y.TimeManager()
    target = 300_000
    t.set_target_frame(target)
    o.timer_target_frame(target)
    rng = np.random.default_rng(9)
    prev, total, lows = t.simulation_step_ns(), 0, set()
    for vmax in rng.uniform(0, 30, 400):
        vmax, d = np.float32(vmax), np.float32(0.01)
        t.on_step_started()
        o.timer_on_step_started()
        total += prev
        assert t.total_simulated_ns == total
        low = min(t.timestep_min_ns, total - target * (total // target))
        lows.add(low < t.timestep_min_ns)
        law = t.law(d)
        assert law.timestep_min_ns == low and law.simulation_step_ns == prev
        cfl = exact_round_ns(np.float32(1.5) * np.float32(0.4) * d / (vmax + np.float32(0.00001)))
        want = max(low, min(min(t.timestep_max_ns, 2 * prev), cfl))
        got = t.update_simulation_step(d, vmax)
        assert got == want == oracle_lib.orc_timer_update(o.h, d, vmax)
        prev = got
    assert lows == {True, False}, "both branches of the lower bound must have been taken"


def test_fixed_timer():
    t = y.TimeManager(fixed_ns=1000000)
    assert t.update_simulation_step(np.float32(0.01), np.float32(5.0)) == 1000000


def test_scene_counts_and_properties():
    w = y.FluidParticleWorld()
    pr = w.properties()
    # world(2.0, 10000, 100): h = 0.02, m = 0.01, radius = 0.005
    assert pr["smoothing_length"] == np.float32(0.02) and pr["particle_mass"] == np.float32(0.01)
    assert pr["particle_radius"] == np.float32(0.005)
    w.reset_fluid(1.0)
    assert w.num_dynamic_particles == 45 * 90 == 4050  # main.rs:181 -> fluidparticleworld.rs:143-146
    assert w.num_boundary_particles == 6840
    p = w.positions
    step = np.float32(0.5) / np.float32(45)
    # jitter in [0.5, 1.0) * step * 0.05 on top of the lattice (fluidparticleworld.rs:156-164)
    lattice = np.stack(np.meshgrid(np.arange(45, dtype=np.float32), np.arange(90, dtype=np.float32)), -1).reshape(-1, 2) * step
    off = p - (np.array([0.1, 0.7], np.float32) + lattice)
    assert (off >= 0.5 * step * 0.05 * 0.999).all() and (off <= step * 0.05 * 1.001).all()
    assert (w.velocities == 0).all()
    w2 = y.FluidParticleWorld()
    w2.reset_fluid(1.0)
    np.testing.assert_array_equal(w2.positions, p)  # deterministic generator


def test_scene_scaling():
    w = y.FluidParticleWorld()
    s = float(np.sqrt(40000 / 4050))
    w.reset_fluid(s)
    n = w.num_dynamic_particles
    assert abs(n - 40000) / 40000 < 0.02
    b = w.boundary_particles
    assert b[:, 0].min() < -1.9 * s and b[:, 0].max() > 3.9 * s  # floor line from -2s to 4s (main.rs:192)


def test_bench_world_counts():
    from util import bench_world

    pos, boundary = bench_world()
    assert len(pos) == 8100  # update_densities.rs:78
    assert 4300 <= len(boundary) <= 4500  # "~4420", SURVEY §3.5


def test_properties_match_oracle():
    from oracle.oracle import Oracle

    o = Oracle().properties()
    w = y.FluidParticleWorld().properties()
    for k in o:
        assert o[k] == w[k], k


def test_sort9_network_is_a_sorting_network():
    """The 25-comparator network hard-coded in csrc/sphx_kernels.hip (sort9), checked with the 0-1 principle."""
    import os
    import re

    src = open(os.path.join(os.path.dirname(y.__file__), "csrc", "sphx_kernels.hip")).read()
    body = src[src.index("void sort9("):]
    body = body[:body.index("}\n")]
    net = [(int(a), int(b)) for a, b in re.findall(r"SPHX_CE\(c(\d), c(\d)\)", body)]
    assert len(net) == 25
    for bits in itertools.product([0, 1], repeat=9):
        a = list(bits)
        for i, j in net:
            if a[i] > a[j]:
                a[i], a[j] = a[j], a[i]
        assert a == sorted(a)


def test_sort9_monotone_network_sorts_every_box_with_ascending_rows_and_columns():
    """The 7-comparator network of csrc/sphx_kernels.hip (sort9_monotone), used for a 3 x 3 cell box inside one 64 x 64 block, whose
    table slots ascend along dx and along dy: the twenty 0-1 matrices with ascending rows and columns (the restricted 0-1 principle),
    and random integer matrices of that shape — among them real boxes: slot = y bits | x bits of interleaved coordinates."""
    import os
    import re

    src = open(os.path.join(os.path.dirname(y.__file__), "csrc", "sphx_kernels.hip")).read()
    body = src[src.index("void sort9_monotone("):]
    body = body[:body.index("}\n")]
    net = [(int(a), int(b)) for a, b in re.findall(r"SPHX_CE\(c(\d), c(\d)\)", body)]
    assert len(net) == 7

    def run(a):
        a = list(a)
        for i, j in net:
            if a[i] > a[j]:
                a[i], a[j] = a[j], a[i]
        return a

    count = 0
    for bits in itertools.product([0, 1], repeat=9):
        m = np.array(bits).reshape(3, 3)
        if (np.diff(m, axis=0) >= 0).all() and (np.diff(m, axis=1) >= 0).all():
            count += 1
            assert run(bits) == sorted(bits)
    assert count == 20
    rng = np.random.default_rng(5)
    for _ in range(2000):
        m = np.sort(np.sort(rng.integers(0, 1000, (3, 3)), axis=0), axis=1)
        assert run(m.reshape(-1)) == sorted(m.reshape(-1))

    def spread(v):  # bit k of v -> bit 2 k
        return sum(((v >> k) & 1) << (2 * k) for k in range(6))

    for cx in range(1, 63):
        for cy in range(1, 63, 7):
            slots = [(spread(cy + dy) << 1) | spread(cx + dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
            assert run(slots) == sorted(slots)


def test_block_mapping_is_a_permutation_in_both_sweep_directions():
    """xcd_bid(rev, shift) (sphx_kernels.hip): workgroup b of a grid of G (a multiple of 8) belongs to XCD x = b mod 8 and is that
    XCD's q-th workgroup, q = b / 8 bottom-up or G / 8 - 1 - b / 8 top-down (round 6: consecutive launches alternate).  XCD x takes
    every eighth chunk of 2^shift consecutive particle blocks (and an eighth of the short last group); shift 0: one contiguous eighth
    (rounds 1-5).  Every block is taken exactly once, each XCD's blocks ascend with q, top-down is bottom-up reversed."""

    def blk(b, g, rev, sh):
        per, q, x = g // 8, b >> 3, b & 7
        if rev:
            q = per - 1 - q
        if sh == 0:
            return x * per + q
        full, grp = per >> sh, q >> sh
        if grp < full:
            return (grp << (sh + 3)) + (x << sh) + (q - (grp << sh))
        r = per - (full << sh)
        return (full << (sh + 3)) + x * r + (q - (full << sh))

    for g in (8, 16, 64, 3912, 4096, 4104, 62504, 500000):
        for sh in (0, 4, 6, 9):
            per = g // 8
            fwd = [blk(b, g, False, sh) for b in range(g)]
            rev = [blk(b, g, True, sh) for b in range(g)]
            assert sorted(fwd) == list(range(g)) and sorted(rev) == list(range(g))
            for x in range(8):
                f = [v for b, v in enumerate(fwd) if b & 7 == x]
                r = [v for b, v in enumerate(rev) if b & 7 == x]
                assert f == sorted(f) and r == f[::-1]
                if sh == 0 or per <= (1 << sh):
                    assert f == list(range(x * per, (x + 1) * per))
