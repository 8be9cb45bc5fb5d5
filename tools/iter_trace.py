#!/usr/bin/env python3
"""Where does the dam-break leave the Id = Iv = 1 regime?  Runs the bench scene for --steps steps and prints, per block of --every
steps, the mean/max solver iteration counts, warm-start rates, dt and ms/step.  (GPU box; product path only.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yasph2d_amd as y  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=4000)
ap.add_argument("--every", type=int, default=100)
ap.add_argument("--fixed", type=int, nargs=2, default=(0, 0))
args = ap.parse_args()

scale = float(np.sqrt(args.particles / 4050.0))
w = y.FluidParticleWorld()
w.reset_fluid(scale)
pos, boundary = w.positions, w.boundary_particles
diam = np.float32(2.0) * np.float32(w.properties()["particle_radius"])
timer = y.TimeManager()
ctx = y.SphxContext(y.default_params(fixed_iterations=tuple(args.fixed)))
ctx.set_boundary(boundary)
ctx.upload(pos)
rows = []
blk = []
ctx.synchronize()
t0 = time.perf_counter()
for s in range(args.steps):
    vmax = ctx.step_begin(timer.simulation_step(), timer.law(diam))
    dt_ns = timer.update_simulation_step(diam, vmax)
    st = ctx.step_finish(y.duration_as_secs_f32(dt_ns))
    blk.append(st)
    if len(blk) == args.every:
        ctx.synchronize()
        t1 = time.perf_counter()
        rows.append(dict(step=s + 1, ms_per_step=(t1 - t0) / len(blk) * 1e3, Id=float(np.mean([b["density_iterations"] for b in blk])),
                         Id_max=int(max(b["density_iterations"] for b in blk)), Iv=float(np.mean([b["divergence_iterations"] for b in blk])),
                         Iv_max=int(max(b["divergence_iterations"] for b in blk)), Wd=float(np.mean([b["warmstart_density"] for b in blk])),
                         Wv=float(np.mean([b["warmstart_divergence"] for b in blk])), dt=blk[-1]["dt"], vmax=blk[-1]["vmax"],
                         kbar=blk[-1]["neighbor_entries"] / len(pos), flags=int(np.bitwise_or.reduce([b["flags"] for b in blk]))))
        print(json.dumps(rows[-1]), flush=True)
        blk = []
        t0 = time.perf_counter()
