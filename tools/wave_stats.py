#!/usr/bin/env python3
"""List-length statistics per wavefront after K steps of the dam break (GPU): how many wavefronts have a list past the staged rows.
   tools/wave_stats.py PARTICLES STEPS"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import yasph2d_amd as y

n_target, steps = int(sys.argv[1]), int(sys.argv[2])
w = y.FluidParticleWorld()
w.reset_fluid(float(np.sqrt(n_target / 4050.0)))
t = y.TimeManager()
s = y.DFSPHSolver(w, y.default_params())
if steps:
    s.simulation_steps(w, t, steps, sync_world=False)
else:
    s.simulation_step(w, t, sync_world=False)
ctx = s.context()
counts, _, _ = ctx.download_neighbors()
ct = counts[:, 1].astype(np.int64)
W = len(ct) // 64
m = ct[:W * 64].reshape(W, 64).max(1)
print("particles", len(ct), "steps", steps, "mean ct %.2f" % ct.mean(), "wave max mean %.2f" % m.mean(), "hist", np.bincount(m)[:24].tolist())
for lim in (9, 12, 15, 16):
    print("  waves with a list longer than", lim, ": %.1f %%" % (100.0 * (m > lim).mean()))
