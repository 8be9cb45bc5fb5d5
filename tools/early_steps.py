#!/usr/bin/env python3
"""Wall time of single steps of the 16 M scene, un-profiled (one library call per step + a stream synchronisation), around idle gaps:
what bench.py's BusyGpu is there for.  Run on the GPU box from the repository root.
  early_steps.py prewarm [scratch particles, 0 = none] [ms]   the first 70 steps from t = 0 behind a scratch-context warm-up that ENDS
                                                              before the measured context uploads (bench.py until round 6)
  early_steps.py gap                                          60 steps, then 5 / 30 / 300 ms of host sleep, 30 more steps each
  early_steps.py busy [scratch particles]                     a scratch context keeps the GPU busy THROUGH the measured context's
                                                              first step (the upload): the steps behind it (bench.py's BusyGpu)"""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
import yasph2d_amd as y

def world_of(n):
    w = y.FluidParticleWorld(); w.reset_fluid(float(np.sqrt(n / 4050.0))); return w

def timed(solver, w, tm, ctx, k, sync_first=True):
    ts = []
    for _ in range(k):
        if sync_first: ctx.synchronize()
        t0 = time.perf_counter()
        solver.simulation_step(w, tm, sync_world=False)
        ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    return ts

mode = sys.argv[1] if len(sys.argv) > 1 else "gap"
w = world_of(16.0e6)
solver = y.DFSPHSolver(w); tm = y.TimeManager(); ctx = solver.context()
if mode == "prewarm":
    pw_n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
    pw_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
    if pw_n:
        sw = world_of(pw_n); ss = y.DFSPHSolver(sw); st = y.TimeManager()
        t_end = time.perf_counter() + pw_ms * 1e-3
        while time.perf_counter() < t_end:
            ss.simulation_steps(sw, st, 10, sync_world=False)
        ss.context().synchronize(); ss.close()
    ts = timed(solver, w, tm, ctx, 70)
    print(f"prewarm {pw_n} particles {pw_ms} ms, then the upload:", " ".join(f"{t:.0f}" for t in ts[1:13]),
          "| mean of steps 5-24: %.0f  25-44: %.0f  45-69: %.0f" % (np.mean(ts[5:25]), np.mean(ts[25:45]), np.mean(ts[45:70])))
elif mode == "gap":
    a = timed(solver, w, tm, ctx, 60)
    print("from t = 0 (behind the upload):", " ".join(f"{t:.0f}" for t in a[1:16]), "| steps 45-59: %.0f" % np.mean(a[45:]))
    for gap in (0.005, 0.03, 0.3):
        time.sleep(gap)
        b = timed(solver, w, tm, ctx, 30)
        print(f"after {gap * 1e3:.0f} ms of idling:", " ".join(f"{t:.0f}" for t in b[:15]), "| steps 20-29: %.0f" % np.mean(b[20:]))
else:
    scr_n = float(sys.argv[2]) if len(sys.argv) > 2 else 16e6
    sw = world_of(scr_n); ss = y.DFSPHSolver(sw); st = y.TimeManager()
    ss.simulation_steps(sw, st, 2, sync_world=False)
    stop = threading.Event()
    def loop():
        while not stop.is_set():
            ss.simulation_steps(sw, st, 4, sync_world=False)
    th = threading.Thread(target=loop); th.start()
    time.sleep(0.15)
    solver.simulation_step(w, tm, sync_world=False)  # the step that uploads: the scratch context keeps the GPU busy
    stop.set()
    ts = []
    for i in range(45):
        t0 = time.perf_counter()
        solver.simulation_step(w, tm, sync_world=False)
        ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        if i == 3: th.join()
    print(f"busy through the upload (scratch context of {scr_n:.0f} particles):", " ".join(f"{t:.0f}" for t in ts[:16]),
          "| mean of steps 5-24: %.0f  25-44: %.0f" % (np.mean(ts[4:24]), np.mean(ts[24:44])))
