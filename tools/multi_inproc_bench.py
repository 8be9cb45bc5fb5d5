#!/usr/bin/env python3
"""ms/step of sphx_multi with N tiles in ONE process on device 0 (functional timing of the in-process tile driver)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yasph2d_amd as y
from yasph2d_amd.multi import MultiSolver
world, per, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
w = y.FluidParticleWorld(); w.reset_fluid(float(np.sqrt(per * world / 4050.0)))
m = MultiSolver(y.default_params(), devices=[0] * world)
m.set_boundary(w.boundary_particles); m.upload(w.positions)
t = y.TimeManager()
for _ in range(10): m.step(t)
m.synchronize(); t0 = time.perf_counter()
for _ in range(steps): m.step(t)
m.synchronize(); el = time.perf_counter() - t0
print(f"world {world} x {per}: {el / steps * 1e3:.4f} ms/step, {len(w.positions) * steps / el / 1e9:.3f} G particle-steps/s, overlap={os.environ.get('SPHX_MULTI_NO_OVERLAP') != '1'}")
