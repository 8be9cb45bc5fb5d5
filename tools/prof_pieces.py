"""Time the pieces FluidParticleWorld exposes (update_neighborhood, update_densities, compute_alpha) at 1M particles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import yasph2d_amd as y
w = y.FluidParticleWorld(); w.reset_fluid(float(np.sqrt(1e6 / 4050)))
ctx = y.SphxContext(); ctx.set_boundary(w.boundary_particles); ctx.upload(w.positions)
timer = y.TimeManager()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    vmax = ctx.step_begin(timer.simulation_step()); ctx.step_finish(y.duration_as_secs_f32(timer.update_simulation_step(np.float32(0.01), vmax)))
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(20):
    ctx.update_neighborhood(); ctx.update_densities(); ctx.compute_alpha()
ctx.profile_enable(False)
for k, v in sorted(ctx.profile_get().items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{k:40s} {v['total_ms'] / v['launches'] * 1000:8.1f} us x{v['launches']}")
