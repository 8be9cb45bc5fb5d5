#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference cannot be built or imported in this image (Rust, no toolchain), and it ships no golden vectors for the
DFSPH step.  The fixtures are therefore outputs of the oracle (oracle/sph_oracle.cpp) on inputs produced by the host
mirror's scene helpers; they pin (a) the oracle against accidental change and (b) the HIP path against a committed
answer that does not need the oracle at run time.  What IS pinned by the reference itself (Morton known answers,
neighbour-search and kernel properties) is tested directly in tests/test_oracle_*.py.

Files (numpy .npz, a few hundred KB in total):
  dam_break_4050.npz   inputs (pos, boundary) of the main.rs:177-196 scene + oracle state after 1, 10, 100 adaptive steps
  dam_break_4050_fixed.npz  same scene, fixed 3 density / 2 divergence iterations, after 20 steps
  bench_world.npz      benches/benchmarks/update_densities.rs world: sorted order, cells, neighbour lists, densities per kernel
  uniform_1000.npz     the neighbour-search test workload (1000 points, density 10, R = 1): sorted order + lists
  wcsph_dam_break_4050.npz  WCSPH (solver/wscsph.rs, cfl factor 0.2) on the same scene: state after 1, 50, 300 steps
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from util import bench_world, dam_break, uniform_points  # noqa: E402

from oracle.oracle import KERNEL_POLY6, KERNEL_SPIKY, KERNEL_WENDLAND, Oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest()[:8], np.uint64)[0]


def state(o, with_lists=False):
    c, s, l = o.neighbors()
    d = dict(pos=o.positions(), vel=o.velocities(), density=o.densities(), ids=o.ids(), alpha=o.alpha(), kappa=o.kappa(),
             stiffness=o.stiffness(), nb_counts=c, nb_digest=digest(c, l), timer_ns=np.uint64(o.timer_step_ns()))
    if with_lists:
        d["nb_lists"] = l
    return d


def dam(fixed, checkpoints, name):
    pos, boundary = dam_break(1.0)
    o = Oracle()
    o.set_fixed_iterations(*fixed)
    o.set_boundary(boundary)
    o.set_particles(pos)
    out = dict(in_pos=pos, in_boundary=boundary, fixed=np.array(fixed, np.uint32))
    stats = []
    for s in range(1, max(checkpoints) + 1):
        st = o.dfsph_step()
        stats.append([st["density_iterations"], st["divergence_iterations"], st["warmstart_density"], st["warmstart_divergence"]])
        if s in checkpoints:
            for k, v in state(o).items():
                out[f"s{s}_{k}"] = v
    out["iterations"] = np.array(stats, np.uint32)
    np.savez_compressed(os.path.join(OUT, name), **out)


def wcsph(checkpoints, name):
    import yasph2d_amd as y

    pos, boundary = dam_break(1.0)
    t = y.TimeManager(cfl_factor=0.2)
    o = Oracle()
    o.timer_adaptive(t.timestep_max_ns, t.timestep_min_ns, 0.2)
    o.set_boundary(boundary)
    o.set_particles(pos)
    out = dict(in_pos=pos, in_boundary=boundary, timestep_max_ns=np.uint64(t.timestep_max_ns), timestep_min_ns=np.uint64(t.timestep_min_ns))
    for s in range(1, max(checkpoints) + 1):
        o.wcsph_step()
        if s in checkpoints:
            c, _, l = o.neighbors()
            out.update({f"s{s}_pos": o.positions(), f"s{s}_vel": o.velocities(), f"s{s}_density": o.densities(), f"s{s}_ids": o.ids(),
                        f"s{s}_nb_digest": digest(c, l), f"s{s}_timer_ns": np.uint64(o.timer_step_ns())})
    np.savez_compressed(os.path.join(OUT, name), **out)


def main():
    wcsph((1, 50, 300), "wcsph_dam_break_4050.npz")
    dam((0, 0), (1, 10, 100), "dam_break_4050.npz")
    dam((3, 2), (20,), "dam_break_4050_fixed.npz")

    pos, boundary = bench_world()
    o = Oracle()
    o.set_boundary(boundary)
    o.set_particles(pos)
    o.update_neighborhood()
    out = dict(in_pos=pos, in_boundary=boundary, ids=o.ids(), boundary_ids=o.boundary_ids())
    c, s, l = o.neighbors()
    out.update(nb_counts=c, nb_lists=l)
    for static in (False, True):
        f, ci = o.cells(static)
        out[f"cells_first_{int(static)}"] = f
        out[f"cells_cidx_{int(static)}"] = ci
    for kind, nm in ((KERNEL_WENDLAND, "wendland"), (KERNEL_POLY6, "poly6"), (KERNEL_SPIKY, "spiky")):
        o.update_densities(kind)
        out[f"density_{nm}"] = o.densities()
    o.compute_alpha()
    out["alpha"] = o.alpha()
    np.savez_compressed(os.path.join(OUT, "bench_world.npz"), **out)

    pos = uniform_points(1000, 10.0, 123456789)
    o = Oracle(search_radius=1.0)
    o.set_particles(pos)
    o.update_neighborhood()
    c, s, l = o.neighbors()
    f, ci = o.cells()
    np.savez_compressed(os.path.join(OUT, "uniform_1000.npz"), in_pos=pos, ids=o.ids(), nb_counts=c, nb_lists=l, cells_first=f, cells_cidx=ci)
    for n in sorted(os.listdir(OUT)):
        if n.endswith(".npz"):
            print(n, os.path.getsize(os.path.join(OUT, n)))


if __name__ == "__main__":
    main()
