#!/usr/bin/env python3
"""bench.py — particle-steps/sec of the 2D DFSPH dam-break (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--particles P]

`--gpus N` with N > 1 started WITHOUT a launcher (no RANK in the environment) starts the N ranks itself: before anything touches
the GPU this process runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py
<same arguments>` as a CHILD process (never an exec of a GPU-touching process), forwards rank 0's one JSON line and exits with the
child's status.  Started under a launcher (the driver's `python -m torch.distributed.run ... bench.py --gpus N`), WORLD_SIZE must
equal --gpus or the run fails loudly.  One process per GPU; RCCL ("nccl") carries the halo records.

A "step" is one Solver::simulation_step (dfsph.rs:414-525) of the dam-break scene of main.rs:177-196 scaled to ~P
particles per GPU, driven exactly like the drop-in shim would drive it: sphx_step_begin -> host TimeManager CFL law ->
sphx_step_finish, adaptive timer from t = 0.  Inputs are uploaded before the timed region starts (device resident).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0  # what a float4 copy reaches on it (same guide): the practical ceiling of a streaming kernel
SIMDS = 256 * 4              # 256 CUs x 4 SIMDs (same guide); a wave64 vector instruction occupies its SIMD for 2, 4 or 8 cycles (tools/valu_issue_bench.hip)
ENGINE_CLOCK_HZ = 2.4e9      # peak engine clock


def bytes_per_particle_step(kbar, Id, Iv, Wd, Wv, compressed=False, rbar=0.5, folded=0.0):
    """SURVEY.md §8(d) / BASELINE.md §4 list-based algorithmic bytes per particle-step.  compressed: the workgroup-local 10-bit
    list format of this build — every list-consuming traversal moves the 16-bit count word (+ one format word per wavefront), 4/3*kbar of entries (three to a 32-bit word)
    and 4*rbar of out-of-window table lines (rbar = such entries per particle, ~0.5) instead of the 8 + 4*kbar of the 32-bit lists.
    folded: list traversals per step that the neighbour build does while it still holds the list in registers (the divergence loop's
    first compute_density_change, or its warm start: SPHX_FUSE_DIV) — their list read does not happen and is not counted."""
    lst = (2 + 4.0 / 64 + (8.0 / 6.0) * kbar + 4 * rbar + 4.0 / 256) if compressed else (8 + 4 * kbar)
    if compressed:
        save = (8 + 4 * kbar) - lst  # per traversal
        return (252 + 16 * kbar - 4 * save + Id * (84 + 8 * kbar - 2 * save) + Iv * (80 + 8 * kbar - 2 * save) + (Wd + Wv) * (44 + 4 * kbar - save)
                - folded * lst)
    return 252 + 16 * kbar + Id * (84 + 8 * kbar) + Iv * (80 + 8 * kbar) + (Wd + Wv) * (44 + 4 * kbar) - folded * lst


def bytes_per_particle_step_this_build(kbar, Id, Iv, Wd, Wv, rbar, compressed=True, fuse_div=True, fuse_predict=True):
    """Per-particle-step bytes of THIS build's arrays (each per-particle array a pass touches, once; neighbour records
    assumed cache-served) — smaller than the SURVEY.md §8(d) model above, which prices the reference's array set: positions and
    velocities are separate 8-byte arrays (a velocity-only pass moves 8 bytes, not 16), the first correction of a loop neither reads
    nor zeroes the warm-start value, the divergence loop's first compute_density_change (or its warm start) rides on the neighbour
    build, the velocity prediction on the density loop's first compute_density_error when no warm start precedes it (the acceleration
    read and the velocity written there: 16 bytes instead of a pass of 24), and a list is 2 + 4/3 k + 4 r bytes."""
    L = (2 + 4.0 / 64 + (8.0 / 6.0) * kbar + 4 * rbar + 4.0 / 256) if compressed else (8 + 4 * kbar)
    nonpressure = 16 + 4 + L + 8
    predict = (24 * min(Wd, 1.0) + 16 * (1.0 - min(Wd, 1.0))) if fuse_predict else 24
    dens_iter_first = (16 + 4 + 4 + L + 4) + (16 + 12 + L + 8 + 4 + 4)   # compute_error + correction (+ the re-grid's cell count: one word)
    dens_iter_more = (16 + 4 + 4 + L + 4) + (16 + 12 + L + 8 + 8 + 4)
    regrid = 6 + 10 + 44                                                  # scan (per particle, dam-break table), scatter, gather (round 5: 6 + 20 + 50 before)
    build = 8 + 8 + L + 4 + 4 + 4                                         # window of positions + velocities, lists out, density, alpha, k / velocity
    div_first = (0 if fuse_div else (16 + 4 + L + 4)) + (16 + 12 + L + 8 + 4)
    div_more = (16 + 4 + L + 4) + (16 + 12 + L + 8 + 8)
    warm_d = 16 + 8 + 8 + 4 + L
    warm_v = (4 + 8) if fuse_div else (16 + 8 + 8 + 4 + L)                # on the build: the stiffness window + the velocity write
    b = nonpressure + predict + regrid + build
    b += dens_iter_first + max(Id - 1.0, 0.0) * dens_iter_more + Wd * warm_d
    b += div_first + max(Iv - 1.0, 0.0) * div_more + Wv * warm_v
    return b


def kernel_source_sha256():
    """Identity of the device code a counter file was taken from: sha256 over the three sources libsphx's kernels are compiled from.
    roofline.traffic / roofline.valu only quote a profiles/ record whose kernels are the ones being benchmarked."""
    import hashlib

    h = hashlib.sha256()
    for f in ("sphx_kernels.hip", "sphx_launch.inc", "sphx_internal.hpp", "sphx_sqrt.hpp"):
        try:
            h.update(open(os.path.join(ROOT, "yasph2d_amd", "csrc", f), "rb").read())
        except OSError:
            return None
    return h.hexdigest()


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


CPU_BASELINE_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
import yasph2d_amd as y            # host-side scene generator only (no GPU call)
from oracle.oracle import Oracle, lib, phase_seconds
scale, budget_s, max_steps = float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
w = y.FluidParticleWorld()
w.reset_fluid(scale)
pos, boundary = np.array(w.positions), np.array(w.boundary_particles)
L = lib(omp=True)
out = {"threads": L.orc_get_max_threads(), "omp_proc_bind": L.orc_get_proc_bind(), "omp_num_places": L.orc_get_num_places(), "n": len(pos)}
for name, allpar in (("port", 0), ("all_parallel", 1)):
    L.orc_set_all_parallel(allpar)
    o = Oracle(omp=True)
    o.set_boundary(boundary)
    o.set_particles(pos)
    o.dfsph_step()                 # includes the warm-up block, like the GPU warm-up steps; first touch of the pooled vectors
    L.orc_phase_reset()
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps:
        o.dfsph_step()
        steps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    out[name] = {"value": len(pos) * steps / el, "steps": steps, "seconds": el, "phase_seconds_per_step": {k: v / steps for k, v in phase_seconds(L).items()}}
    del o
print("CPUBASE " + json.dumps(out))
"""


def cpu_baseline(scale, budget_s=8.0, max_steps=20):
    """The oracle's OpenMP build (the reference's Rayon loops restated; 'port') timed on this host's cores, in a FRESH child process
    whose environment pins the threads before any OpenMP runtime exists (setting OMP_PROC_BIND inside this process came too late:
    `import torch` had already initialised the libgomp the oracle binds to — round-2 advisor finding).  Two variants: the
    reference-faithful one (`value`: rayon loops parallel, reference-serial loops serial) and `all_parallel` (every loop
    that carries no order runs in parallel).  What the runtime really did with the threads is reported from the library."""
    import subprocess

    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
    env.pop("OMP_NUM_THREADS", None)
    p = subprocess.run([sys.executable, "-c", CPU_BASELINE_CHILD, ROOT, repr(scale), repr(budget_s), str(max_steps)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=600)
    line = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("CPUBASE ")]
    if p.returncode != 0 or not line:
        return {"value": None, "unit": "particle-steps/s", "cores": None, "kind": "port", "error": p.stderr.decode(errors="replace")[-400:]}
    r = json.loads(line[-1][8:])
    bind = {0: "false", 1: "true", 2: "master", 3: "close", 4: "spread"}.get(r["omp_proc_bind"], str(r["omp_proc_bind"]))
    return {
        "value": r["port"]["value"],
        "unit": "particle-steps/s",
        "cores": r["threads"],
        "kind": "port",
        "cpu": cpu_model(),
        "phase_seconds_per_step": r["port"]["phase_seconds_per_step"],
        "phase_note": "wall-clock seconds per step of the restatement's phases; '(serial)' = a loop the reference leaves serial (SURVEY.md 3.1), "
                      "run on one thread in `value` and in parallel in all_parallel",
        "all_parallel": {"value": r["all_parallel"]["value"], "unit": "particle-steps/s", "cores": r["threads"],
                         "phase_seconds_per_step": r["all_parallel"]["phase_seconds_per_step"],
                         "what": "the same restatement with the loops the reference leaves serial (cell indices, apply_sorting, max-velocity scan, "
                                 "velocity prediction, warm-start clamps) also in parallel — SURVEY 8(d)'s optional variant",
                         "sample": f"{r['all_parallel']['steps']} steps, {r['all_parallel']['seconds']:.1f} s"},
        "threads": f"OpenMP in a fresh child process: omp_get_max_threads() = {r['threads']}, omp_get_proc_bind() = {bind}, "
                   f"omp_get_num_places() = {r['omp_num_places']} (environment of the child: OMP_PROC_BIND=close OMP_PLACES=cores)",
        "sample": f"{r['port']['steps']} DFSPH steps of the {r['n']}-particle dam-break (same scene generator as the GPU run) after 1 warm-up step, C++/OpenMP restatement of "
                  f"yasph2d's Rayon path (not the Rust binary; per-step vectors pooled like scratch_buffer.rs), {r['port']['seconds']:.1f} s",
    }


def dry_workload(per_gpu, world):
    """The workload string of a run without building the scene (add_fluid_rect, fluidparticleworld.rs:140-166: 90 particles per metre)."""
    sc = float(np.sqrt(per_gpu * world / 4050.0))
    n = int(np.float32(0.5 * sc) * np.float32(90.0)) * int(np.float32(1.0 * sc) * np.float32(90.0))
    name = {4: " = BASELINE configs[3] (64 M, 2x2 tiles)", 8: " = BASELINE configs[4] (128 M, 8 strips)"}.get(world, "") if abs(per_gpu - 16_000_000) < 300_000 else ""
    return f"DFSPH 2D dam-break (main.rs:177-196 scene x{sc:.2f}), ~{n} fluid particles in total (~{n // world} per GPU) on {world} GPU(s){name}"


def launch_ranks(args, argv):
    """`bench.py --gpus N` without a launcher: start N ranks (one per GPU) as a CHILD process tree and forward rank 0's line.
    Nothing in this process has touched the GPU (torch.cuda.device_count() does not initialise HIP on this image)."""
    import socket
    import subprocess

    if args.backend == "nccl" and not args.dry_run:
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible (one rank per GPU over RCCL; "
                             f"--backend gloo lets ranks share a GPU for functional tests)\n")
            return 2
    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL / device-buffer sharing across processes needs it on this pool
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run failed (exit {proc.returncode}, {len(lines)} result lines); command: {' '.join(cmd)}\n")
        return proc.returncode or 1
    out = json.loads(lines[-1])
    if out.get("n_gpus") != args.gpus:
        sys.stderr.write(f"bench.py: asked for {args.gpus} ranks but the result line reports n_gpus = {out.get('n_gpus')}\n")
        return 1
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()
    return 0


def main():
    # dmabuf IPC (RCCL / device-buffer sharing across processes needs it on this pool): must be in the environment before anything
    # loads the HIP runtime — also when an external launcher (the driver's `python -m torch.distributed.run ... bench.py`) started us
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t_process_start = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--skip-steps", type=int, default=0,
                    help="untimed steps before the warm-up (SURVEY 8(d): a second window after 2000 steps = the violent phase, high Id/Iv)")
    ap.add_argument("--particles", type=int, default=None,
                    help="target fluid particles per GPU; default: 16 M (one GPU = BASELINE configs[2], the largest single-GPU config, where the HBM "
                         "roofline binds; 4 GPUs = configs[3], 64 M on 2x2 tiles; 8 GPUs = configs[4], 128 M on 8 strips)")
    ap.add_argument("--no-also", action="store_true", help="one GPU, default workload: skip the two further windows of the line's `also` list "
                                                           "(1 M from t=0 = BASELINE configs[1]; 1 M after 3750 steps)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--halo", type=int, default=16, help="widest ghost halo in cells (multi-GPU); the band in use adapts to the ring budget")
    ap.add_argument("--fixed-halo", action="store_true", help="always exchange the full --halo band")
    ap.add_argument("--tiles", default="auto", choices=["auto", "strips", "grid"],
                    help="multi-GPU layout: strips along the longer side, or a 2 x N/2 grid (auto: 2x2 on 4 GPUs, strips otherwise; SURVEY 8(e))")
    ap.add_argument("--rebalance-every", type=int, default=16, help="steps between re-partitions of the tile cuts (0 = never)")
    ap.add_argument("--comm", default="auto", choices=["auto", "rccl", "torch"],
                    help="halo records: libsphx's own grouped ncclSend/ncclRecv (rccl; auto with --backend nccl) or torch.distributed "
                         "through the sphx_comm_ops table (torch; always with gloo)")
    ap.add_argument("--overlap-exchange", action="store_true",
                    help="halo records on a second stream while the tile counts the cells of the particles it kept (multi-GPU)")
    ap.add_argument("--scalar-comm", default="shm", choices=["shm", "torch"],
                    help="per-step scalar all-reduces: shared-memory (one node) or torch.distributed")
    ap.add_argument("--force-tiles", action="store_true", help="drive a single GPU through the tile driver (profiling the multi-GPU code path)")
    ap.add_argument("--lists-32bit", action="store_true", help="disable the 10-bit neighbour-list compression (A/B runs)")
    ap.add_argument("--solver", default="dfsph", choices=["dfsph", "wcsph"],
                    help="wcsph: the second Solver of the reference (solver/wscsph.rs, cfl factor 0.2, main.rs:116-119) on one GPU")
    ap.add_argument("--no-device-dt", action="store_true", help="plain sphx_step_begin: the device waits for the host's dt (A/B runs)")
    ap.add_argument("--fixed-iterations", type=int, nargs=2, default=(0, 0), metavar=("ID", "IV"),
                    help="run exactly ID constant-density and IV divergence iterations per step (0 0 = adaptive, the reference's behaviour)")
    ap.add_argument("--tolerance-scale", type=float, default=1.0,
                    help="multiply both solver tolerances (dfsph.rs:49,53) by this; < 1 makes the loops iterate (the iterating-regime window)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="a scratch context keeps the GPU busy from this long before the measured context's first warm-up step until that step has "
                         "returned (no idle gap in front of the timed region: BusyGpu; 0 = off)")
    ap.add_argument("--per-step-calls", action="store_true", help="timed region: one library call per step from Python instead of one call "
                    "that runs the K steps (sphx_solver_simulation_steps: the caller's frame loop, main.rs:348-350, inside the library)")
    ap.add_argument("--abi-calls", action="store_true", help="drive the two-phase C ABI from Python (one ctypes call per phase) instead of "
                                                             "the host mirror's one call per step (A/B runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dry-run-scene", action="store_true",
                    help="with --dry-run: every rank also builds the GLOBAL scene on its host, like a real run does before it cuts its tile "
                         "(times the set-up of an N-rank run without a GPU)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher self-test: ranks rendezvous (gloo), barrier, all-reduce and print the line with value = null; no GPU, no compute")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    default_workload = args.particles is None
    if args.particles is None:
        args.particles = 16_000_000
    # the further windows ride on the plain `python bench.py [--steps K --warmup W]` run only (what the driver starts)
    args.also = (default_workload and not args.no_also and args.gpus == 1 and args.solver == "dfsph" and not args.skip_steps
                 and not any(args.fixed_iterations) and args.tolerance_scale == 1.0 and not args.force_tiles and not args.lists_32bit
                 and not args.no_roofline and not args.dry_run)
    launched = "RANK" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {env_world}: start it as `python bench.py --gpus {args.gpus} ...` (it launches "
                 f"its own ranks) or as `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                 f"--master-port P bench.py --gpus {args.gpus} ...`")

    # Exactly ONE line on stdout: RCCL prints a version banner and gloo its connection notes to fd 1, from C code.  Everything
    # written to fd 1 during the run goes to stderr; the result line is written to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
            dist.barrier()
        t0 = time.perf_counter()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        scene_s, scene_n = None, None
        if args.dry_run_scene:
            import yasph2d_amd as y  # host-side scene generator only (no GPU call)

            ts = time.perf_counter()
            wsc = y.FluidParticleWorld()
            wsc.reset_fluid(float(np.sqrt(args.particles * world / 4050.0)))
            scene_n = len(wsc.positions)
            scene_s = time.perf_counter() - ts
            del wsc
            ts_t = torch.tensor([scene_s], dtype=torch.float64)
            if world > 1:
                dist.all_reduce(ts_t, op=dist.ReduceOp.MAX)
            scene_s = float(ts_t.item())
        ipc_env = [os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")]
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seen = torch.tensor([1.0], dtype=torch.float64)
            dist.all_reduce(seen)  # every rank arrived
            assert int(seen.item()) == world
            gathered_env = [None] * world
            dist.all_gather_object(gathered_env, ipc_env[0])  # what every rank's environment holds (set at the top of main())
            ipc_env = gathered_env
        if rank == 0:
            os.write(result_fd, (json.dumps({"metric": "particle-steps/sec (whole node), 2D DFSPH dam-break", "value": None,
                                             "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                             "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                                             "dtype": "f32", "data": "none (dry run of the rank launcher)",
                                             "config": {"workload": "dry run: no compute; would run " + dry_workload(args.particles, world),
                                                        "particles_per_gpu": args.particles,
                                                        "HSA_ENABLE_IPC_MODE_LEGACY_per_rank": ipc_env,
                                                        "scene_build_seconds_max_over_ranks": scene_s, "scene_particles": scene_n,
                                                        "seconds_since_process_start": time.perf_counter() - t_process_start,
                                                        "launcher": "external (RANK was in the environment)" if launched else "bench.py's own child"}}) + "\n").encode())
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev
    if world > 1 or (args.force_tiles and "RANK" in os.environ):  # world 1 under torchrun + --force-tiles: exercises the collectives alone
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    else:
        dist = None
        torch.cuda.set_device(dev_index)

    import yasph2d_amd as y

    def scene_of(particles_total):
        scale = float(np.sqrt(particles_total / 4050.0))
        w = y.FluidParticleWorld()
        w.reset_fluid(scale)
        return scale, w

    class BusyGpu:
        """Keeps the GPU busy, on a SCRATCH context with a scene of its own, from before the measured context's first warm-up step
        until that step (the one that uploads) has returned.  Why: after ANY idle gap of a few milliseconds — and the host generating the
        scene, allocating and copying 16 M particles is tens of them — this part's power management over-corrects: the steps behind the
        gap cost up to 20 % more (the third one most), and it takes ~25 steps, 35 ms, until the step time has settled again
        (tools/early_steps.py, profiles/r06_experiments/idle_gap_transient.txt: after 5 / 30 / 300 ms of idling alike; data do not
        matter).  W = 5 warm-up steps are 7 ms, so a line timed behind them sat inside that transient, 3-4 % below the rate every later
        window of the same run shows.  With the GPU never idle between the scratch steps and the measured context's second warm-up
        step the timed region starts settled.  The measured context still does exactly --warmup untimed and --steps timed steps from
        t = 0; the scratch context has stopped (joined, synchronised) before the timed region's first barrier."""

        def __init__(self):
            self.solver = None
            self.thread = None
            self.stop = threading.Event()

        def start(self, particles):
            if args.prewarm_ms <= 0:
                return False
            if self.solver is None:
                # (a scene of the measured size up to 16 M: a light load settles the clocks for a light load)
                _, self.w = scene_of(min(particles, 16_000_000))
                self.solver = y.DFSPHSolver(self.w, y.default_params(device=dev_index))
                self.timer = y.TimeManager()
                self.solver.simulation_steps(self.w, self.timer, 2, sync_world=False)
            self.stop.clear()

            def loop():
                while not self.stop.is_set():
                    self.solver.simulation_steps(self.w, self.timer, 4, sync_world=False)  # (ctypes drops the GIL for the call)

            self.thread = threading.Thread(target=loop, daemon=True)
            self.thread.start()
            time.sleep(args.prewarm_ms * 1e-3)
            return True

        def signal(self):
            self.stop.set()

        def join(self):
            if self.thread is not None:
                self.stop.set()
                self.thread.join()
                self.thread = None
                self.solver.context().synchronize()

        def close(self):
            self.join()
            if self.solver is not None:
                self.solver.close()
                self.solver = None

    busy = BusyGpu()

    def make_params():
        params = y.default_params(device=dev_index, fixed_iterations=tuple(args.fixed_iterations))
        if args.tolerance_scale != 1.0:
            params.max_avg_density_error = float(np.float32(params.max_avg_density_error) * np.float32(args.tolerance_scale))
            params.max_divergence_error = float(np.float32(params.max_divergence_error) * np.float32(args.tolerance_scale))
        if args.lists_32bit:
            params.list_span_limit = y.LISTS_32BIT
        return params

    fuse_div = os.environ.get("SPHX_FUSE_DIV", "1") != "0"

    def traffic_of(name, n, skip=None):
        # HBM traffic of a kernel from the PMC counters: cannot be collected from inside the process; taken from the committed
        # rocprofv3 passes of this same command (profiles/) when the workload matches, else null.
        import glob

        traffic, src = None, None
        sha = kernel_source_sha256()
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_traffic_*.json"))):
            try:
                tj = json.load(open(tf))
                if sha is None or tj.get("kernel_source_sha256") != sha:
                    continue  # counters of another build of the kernels: not this run's traffic
                if tj.get("skip_steps", 0) != (args.skip_steps if skip is None else skip):
                    continue  # counters of another window of the run (lists are longer once the fluid is compressed)
                # (the library's profile records carry 47 characters of a launch label: "...+divergence_warmstart" arrives without its last two)
                key = next((k for k in tj["bytes_per_launch"] if k == name or (len(name) >= 47 and k.startswith(name))), None)
                if abs(tj["workload_particles"] - n) < 0.02 * n and key is not None:
                    traffic, src = tj["bytes_per_launch"][key]["total"], tj["source"] + f" [file {os.path.basename(tf)}" + (
                        f", taken at git {tj['git_head']}]" if "git_head" in tj else "]")
            except (OSError, KeyError, ValueError):
                pass
        return traffic, src

    def valu_of(name, n):
        # vector instructions / issue cycles of a kernel per launch from the SQ counters of the committed rocprofv3 pass of this command
        # (profiles/r0*_valu_*.json, tools/make_valu_json.py) when the workload matches, else None
        import glob

        rec = None
        sha = kernel_source_sha256()
        for vf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_valu_*.json"))):
            try:
                vj = json.load(open(vf))
                if sha is None or vj.get("kernel_source_sha256") != sha:
                    continue  # counters of another build of the kernels
                if abs(vj["workload_particles"] - n) < 0.02 * n and name in vj["per_launch"]:
                    rec = dict(vj["per_launch"][name], source=vj["source"] + f" [file {os.path.basename(vf)}" + (f", taken at git {vj['git_head']}]" if "git_head" in vj else "]"))
            except (OSError, KeyError, ValueError):
                pass
        return rec

    def measure(one_step, k_steps, ctx, barrier, steps, warmup, skip_steps, want_roofline, n, reduce_max=None):
        """--skip-steps + --warmup untimed steps, then EXACTLY `steps` timed ones between two barriers.  Returns the timing, the step
        statistics and the live roofline record of the dominant kernel."""
        # Warm-up: every launch bracketed by hipEvents on the context's stream -> which kernel dominates a step.  (All ranks take the
        # same decision: the kernel mix is the same on every tile.)
        dominant, wprof = None, None
        for _ in range(skip_steps):
            one_step()
        # (BusyGpu: behind --skip-steps untimed steps the GPU has been busy for long enough anyway)
        kept_busy = skip_steps == 0 and warmup > 0 and busy.start(n)
        if want_roofline and warmup > 0:
            ctx.profile_reset()
            ctx.profile_filter(None)
            ctx.profile_enable(True)
        for k in range(warmup):
            one_step()
            if k == 0 and kept_busy:
                busy.signal()  # the upload is over: the remaining warm-up steps keep the GPU busy themselves
        if kept_busy:
            busy.join()
        if want_roofline and warmup > 0:
            ctx.profile_enable(False)
            wprof = ctx.profile_get()
            dominant = max(wprof.items(), key=lambda kv: kv[1]["total_ms"])[0] if wprof else None
        if want_roofline and dominant is None:
            dominant = "neighbor_build+density_alpha" if args.solver == "dfsph" else "neighbor_build"
        # Timed region.  The dominant kernel alone keeps hipEvent records, around every 4th of its launches (sphx_profile_filter; all of
        # them cost 2.6 % of the step rate, every 4th 0.7 %): its launch duration is measured live, over the steps `value` is computed
        # from, on the stream it runs on.
        if dominant is not None:
            ctx.profile_reset()
            ctx.profile_filter(dominant, 4)
            ctx.profile_enable(True)
        stats = []
        barrier()
        t0 = time.perf_counter()
        if k_steps is not None and not args.per_step_calls:
            stats = k_steps(steps)
        else:
            for _ in range(steps):
                stats.append(one_step())
        barrier()
        elapsed = time.perf_counter() - t0
        live, live_from = None, "timed region"
        if dominant is not None:
            ctx.profile_enable(False)
            live = ctx.profile_get().get(dominant)
            ctx.profile_filter(None)
            if (live is None or not live["launches"]) and wprof and dominant in wprof:
                # fewer than 4 timed launches of the dominant kernel (--steps < 4): the event-bracketed warm-up steps stand in
                live, live_from = wprof[dominant], "warm-up steps (the timed region was too short to bracket a launch)"
        if reduce_max is not None:
            elapsed = reduce_max(elapsed)
        roof = None
        if live is not None and live["launches"]:
            ev_ms = ctx.profile_event_overhead()  # what an EMPTY hipEvent bracket measures on this stream, same process
            avg_raw = live["total_ms"] / live["launches"]
            traffic, traffic_src = traffic_of(dominant, n, skip_steps)
            # per-kernel table: a short extra pass with every launch timed, outside the timed region
            ctx.profile_reset()
            ctx.profile_enable(True)
            extra = max(10, min(steps, 30))
            for _ in range(extra):
                one_step()
            ctx.profile_enable(False)
            prof = ctx.profile_get()
            # What a bracket adds to the kernel inside it, measured on this very run: with EVERY launch bracketed the brackets of a step
            # add up to more than the (gap-free) un-bracketed step takes — the difference, spread over the step's launches, is the
            # inflation per bracket (~1.5 us at 1 M).  (Not the empty bracket: the ~4.6 us marker latency an empty bracket shows hides
            # behind the kernel when there is one in between; subtracting it made the figure 7 % shorter than rocprofv3's kernel trace.
            # With this correction the two agree to ~1 %: profiles/r03_stats_*.txt.)
            sum_bracketed = sum(v["total_ms"] for v in prof.values()) / extra
            per_step = sum(v["launches"] for v in prof.values()) / extra
            infl = max(0.0, (sum_bracketed - elapsed * 1e3 / steps) / per_step) if per_step else 0.0
            infl = min(infl, ev_ms)
            avg_ms = max(avg_raw - infl, 1e-6)
            ach = live["bytes"] / live["launches"] / (avg_ms * 1e-3) / 1e9
            roof = {
                # `bound` names what the evidence says limits the dominant kernel (round-5 review, item 6): a kernel far below its HBM roofline
                # is not "hbm"-bound.  frac stays the contract's figure: algorithmic bytes against the HBM peak (`roofline_of_frac`).
                "bound": "hbm" if ach / HBM_PEAK_GBS >= 0.5 else "latency+issue", "roofline_of_frac": "hbm", "kernel": dominant, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "frac_of_achievable_6300": ach / HBM_ACHIEVABLE_GBS,
                "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": avg_ms, "avg_launch_ms_bracketed": avg_raw,
                "bracket_inflation_ms": infl, "empty_event_bracket_ms": ev_ms, "launches": live["launches"],
                "algorithmic_bytes_per_launch": live["bytes"] / live["launches"],
                "measured": "hipEvents around every 4th launch of this kernel inside the timed region, on the context's stream, minus "
                            "bracket_inflation_ms = (sum of all bracketed launches of a step - the un-bracketed step) / launches per step, from "
                            "a pass with every launch bracketed" if live_from == "timed region" else
                            "hipEvents around this kernel's launches, minus bracket_inflation_ms; taken from the " + live_from,
                "per_kernel_ms_per_step_event_inflated": {k: v["total_ms"] / extra for k, v in sorted(prof.items())},
                "note": "per_kernel_ms_per_step_event_inflated brackets EVERY launch with events (which keeps kernels from overlapping their "
                        "neighbours' tails): its sum exceeds ms_per_step; information only",
                "vs_rocprofv3": "avg_launch_ms is taken on the settled GPU the line is timed on (config.prewarm).  The committed rocprofv3 "
                                "passes of this command run WITHOUT the scratch context (its launches would be averaged in): "
                                "profiles/r06_stats_16M.txt averages 22 steps from t = 0, i.e. the transient behind the upload (~8 % above); "
                                "profiles/r06_stats_16M_settled_steps45-64.txt is the same pass over steps 45-64 and agrees",
            }
            # Whole-step ledger per kernel: HBM bytes per launch from the committed counter passes of THESE kernels (hash-gated like
            # `traffic`), launch duration and launches per step live from this run's every-launch pass (event-inflated, see note).
            ledger = {}
            for kname, v in sorted(prof.items()):
                kb, _ = traffic_of(kname, n, skip_steps)
                if not v["launches"]:
                    continue
                us = max(v["total_ms"] / v["launches"] - infl, 1e-6) * 1e3
                ledger[kname] = {"launches_per_step": v["launches"] / extra, "us_per_launch": us,
                                 "hbm_bytes_per_launch": kb, "tb_per_s": (kb / (us * 1e-6) / 1e12) if kb else None,
                                 "algorithmic_bytes_per_launch": v["bytes"] / v["launches"] if v.get("bytes") else None}
            roof["per_kernel"] = ledger
            roof["per_kernel_note"] = ("us_per_launch: hipEvent brackets around every launch of a separate pass outside the timed region, minus "
                                       "bracket_inflation_ms; hbm_bytes_per_launch: FETCH_SIZE (doubled, gfx950) + WRITE_SIZE of the committed "
                                       "rocprofv3 --pmc passes (traffic_source), null when those were taken from other kernel sources")
            if roof["bound"] != "hbm":
                roof["bound_note"] = ("the dominant kernel is co-limited by vector issue (slot occupancy ~0.9, ~0.65 of the SIMD time with the 2-cycle "
                                      "instruction classes counted as such) and by the dependent round trips of a wavefront's life: +9 % vector "
                                      "instructions cost +2.8 % of its time (round 5), -5 500 cycles of waiting per wavefront buy 2 %, fewer "
                                      "instructions in more trips lose 7 % (round 6, profiles/r06_experiments/far_prefetch.txt).  achieved / peak / "
                                      "frac are its algorithmic bytes against the HBM peak, as the bench contract prescribes")
            vrec = valu_of(dominant, n) if dominant.startswith("neighbor_build") else None
            if vrec and vrec.get("active_inst_valu"):
                # The neighbour build is far from its HBM roofline and NOT simply bound by vector issue either (DESIGN.md section 4,
                # profiles/r05_experiments/build_phase_stamps.txt, profiles/r06_experiments/far_prefetch.txt): it sits at the
                # knee between its vector time and the chain of round trips a wavefront waits for (rounds 3-4 said "valu" in `bound`, round 5
                # "hbm"; since round 6 it says "latency+issue" while frac < 0.5).  The issue-slot occupancy is stated beside it — only when the
                # committed SQ counters were taken from THESE kernels (kernel_source_sha256).
                roof["valu_issue_frac"] = vrec["active_inst_valu"] * 4.0 / (SIMDS * ENGINE_CLOCK_HZ * avg_ms * 1e-3)
                roof["valu"] = {"insts_valu_per_wavefront": vrec["insts_valu_per_wave"], "wavefronts": vrec["waves"],
                                "sq_active_inst_valu": vrec["active_inst_valu"], "simds": SIMDS, "engine_clock_hz": ENGINE_CLOCK_HZ,
                                "formula": "SQ_ACTIVE_INST_VALU * 4 cycles / (SIMDs * engine clock * avg_launch_ms): the fraction of every SIMD's "
                                           "issue slots the launch fills, at the 2.4 GHz peak clock (the chip runs 2.1-2.3 GHz under this load, so "
                                           "the real fraction is higher)",
                                "instruction_costs": "measured on this part (tools/valu_issue_bench.hip, profiles/r05_issue_rate_*.txt): a wave64 "
                                                     "instruction occupies its SIMD for 4 cycles (fma, min/max, shifts left, bit-field, compares, selects, "
                                                     "DPP, packed f32), 8 cycles (sqrt/rsq/rcp) or 2 cycles (f32 add/sub/mul, mov, integer add/sub, logic, "
                                                     "right shifts — when a second wavefront co-issues); SQ_ACTIVE_INST_VALU counts ONE 4-cycle slot per "
                                                     "instruction of the 2- and 4-cycle classes and two per 8-cycle one, so this figure is the slot "
                                                     "occupancy, an upper bound of the time the SIMD is busy",
                                "source": vrec["source"]}
        return elapsed, stats, roof

    def iteration_stats(stats):
        return dict(Id=float(np.mean([s["density_iterations"] for s in stats])), Iv=float(np.mean([s["divergence_iterations"] for s in stats])),
                    Wd=float(np.mean([s["warmstart_density"] for s in stats])), Wv=float(np.mean([s["warmstart_divergence"] for s in stats])),
                    Id_max=int(max(s["density_iterations"] for s in stats)), Iv_max=int(max(s["divergence_iterations"] for s in stats)))

    def step_model(kb, rb, it, n, steps, elapsed, measured_k):
        # traversals the neighbour build does on the spot (SPHX_FUSE_DIV): one per step — the divergence loop's first
        # compute_density_change, or that loop's warm start when it has one
        folded = 1.0 if (fuse_div and args.solver == "dfsph") else 0.0
        bstep_ref = bytes_per_particle_step(kb, it["Id"], it["Iv"], it["Wd"], it["Wv"])
        bstep = bytes_per_particle_step(kb, it["Id"], it["Iv"], it["Wd"], it["Wv"], compressed=not args.lists_32bit, rbar=rb, folded=folded)
        blay = bytes_per_particle_step_this_build(kb, it["Id"], it["Iv"], it["Wd"], it["Wv"], rb, compressed=not args.lists_32bit,
                                                   fuse_div=fuse_div and args.solver == "dfsph",
                                                   fuse_predict=os.environ.get("SPHX_FUSE_PREDICT", "1") != "0" and not args.no_device_dt and not args.force_tiles)
        return {"bytes_per_particle_step": bstep, "bytes_per_particle_step_32bit_lists": bstep_ref,
                "bytes_per_particle_step_this_build": blay,
                "bytes_note": "bytes_per_particle_step = SURVEY.md 8(d)'s list-based model of the reference's array set (what frac_of_hbm_peak_whole_step "
                              "is computed from, as the survey prescribes); bytes_per_particle_step_this_build = what this build's own arrays add up to "
                              "(split position/velocity arrays, no warm-start zeroing, folded traversal, 10-bit lists): the HBM rate the device really "
                              "sustains over the step is achieved_GBs_this_build_per_gpu",
                "achieved_GBs_this_build_per_gpu": blay * n * steps / elapsed / 1e9,
                "frac_of_hbm_peak_this_build": blay * n * steps / elapsed / 1e9 / HBM_PEAK_GBS,
                "list_format": "32-bit" if args.lists_32bit else "workgroup-local 10-bit slots, three to a 32-bit word (32-bit fallback per wavefront)",
                "mean_neighbors": kb, "out_of_window_entries_per_particle": rb,
                "k_and_r": "measured: list entries of the latest neighbour build / particles it ran over" if measured_k else "not measured",
                "list_traversals_folded_into_the_neighbour_build_per_step": folded,
                "achieved_GBs_whole_step_per_gpu": bstep * n * steps / elapsed / 1e9,
                "frac_of_hbm_peak_whole_step": bstep * n * steps / elapsed / 1e9 / HBM_PEAK_GBS,
                "frac_of_achievable_6300_whole_step": bstep * n * steps / elapsed / 1e9 / HBM_ACHIEVABLE_GBS,
                "note": "at 1 M particles the working set (~170 MB) sits inside the 256 MB Infinity Cache: the HBM roofline is "
                        "a soft bound there; 16 M (the default) is the size where it binds"}

    def single_window(particles, skip_steps, steps, warmup, want_roofline=True):
        """One measurement on ONE context: the dam-break scaled to `particles`, `skip_steps` untimed steps, warm-up, timed steps."""
        scale, w = scene_of(particles)
        pos, boundary = w.positions, w.boundary_particles
        n = len(pos)
        diam = np.float32(2.0) * np.float32(w.properties()["particle_radius"])
        timer = y.TimeManager(cfl_factor=0.2) if args.solver == "wcsph" else y.TimeManager()
        params = make_params()
        k_steps = None
        if args.no_device_dt or args.abi_calls:
            # the raw two-phase C ABI, one ctypes call per phase (A/B runs)
            solver = None
            ctx = y.SphxContext(params)
            ctx.set_boundary(boundary)
            ctx.upload(pos)

            def one_step():
                if args.solver == "wcsph":
                    vmax = ctx.wcsph_step_begin(timer.simulation_step())
                    return ctx.wcsph_step_finish(y.duration_as_secs_f32(timer.update_simulation_step(diam, vmax)))
                vmax = ctx.step_begin(timer.simulation_step(), None if args.no_device_dt else timer.law(diam))
                dt_ns = timer.update_simulation_step(diam, vmax)
                return ctx.step_finish(y.duration_as_secs_f32(dt_ns))

        else:
            # Solver::simulation_step(&mut world, &mut time_manager) through the host-side mirror of the shim (sph::HipDfsphSolver,
            # csrc/sphx_host.cpp — the C++ twin of the Rust impl in INTEGRATION.md): world, TimeManager and solver object live behind
            # the boundary, one call per step, positions stay device-resident (sync_world = 0).  The arrays are uploaded by the first
            # (warm-up) step, like the reference reads its Vecs.
            solver = (y.WCSPHSolver if args.solver == "wcsph" else y.DFSPHSolver)(w, params)
            ctx = solver.context()

            def one_step():
                return solver.simulation_step(w, timer, sync_world=False)

            def k_steps(k):
                return solver.simulation_steps(w, timer, k, sync_world=False)

        def barrier():
            torch.cuda.synchronize()
            ctx.synchronize()

        elapsed, stats, roof = measure(one_step, k_steps, ctx, barrier, steps, warmup, skip_steps, want_roofline, n)
        it = iteration_stats(stats)
        kbar = float(np.mean([s["neighbor_entries"] for s in stats])) / n
        rbar = float(np.mean([s.get("remote_entries", 0) for s in stats])) / n
        res = dict(scale=scale, n=n, n_boundary=len(boundary), elapsed=elapsed, steps=steps, warmup=warmup, skip_steps=skip_steps, it=it, kbar=kbar, rbar=rbar,
                   roof=roof, value=n * steps / elapsed, ms_per_step=elapsed / steps * 1e3, model=step_model(kbar, rbar, it, n, steps, elapsed, True))
        if solver is not None and hasattr(solver, "close"):
            solver.close()
        elif solver is None:
            ctx.close()
        return res

    def window_summary(r, label):
        """The compact form of a further window for the `also` list of the line."""
        o = {"window": label, "value": r["value"], "unit": "particle-steps/s", "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"],
             "skip_steps": r["skip_steps"], "particles": r["n"], "mean_density_iterations": r["it"]["Id"], "mean_divergence_iterations": r["it"]["Iv"],
             "warmstart_rate": [r["it"]["Wd"], r["it"]["Wv"]], "mean_neighbors": r["kbar"],
             "frac_of_hbm_peak_whole_step": r["model"]["frac_of_hbm_peak_whole_step"], "bytes_per_particle_step": r["model"]["bytes_per_particle_step"]}
        if r["roof"]:
            o["dominant_kernel"] = {k: r["roof"][k] for k in ("kernel", "avg_launch_ms", "achieved", "frac", "algorithmic_bytes_per_launch", "traffic")}
        return o

    config_name = {1: "BASELINE configs[1]", 2: "BASELINE configs[2] (16 M, the largest single-GPU config)", 4: "BASELINE configs[3] (64 M, 2x2 tiles)",
                   8: "BASELINE configs[4] (128 M, 8 strips)"}
    multi = None
    if world == 1 and not args.force_tiles:
        head = single_window(args.particles, args.skip_steps, args.steps, args.warmup, want_roofline=not args.no_roofline)
        also = []
        if args.also:
            # the windows the headline (configs[2], 16 M: the largest single-GPU config) does not show: configs[1] — 1 M particles, resident
            # in the Infinity Cache — and the iterating regime of the reference scene (after 3 750 steps the divergence loop needs two
            # iterations and a warm start)
            also.append(window_summary(single_window(1_000_000, 0, 100, 5), "DFSPH 1 M particles from t=0 (BASELINE configs[1])"))
            also.append(window_summary(single_window(1_000_000, 3750, 100, 5), "DFSPH 1 M particles after 3750 steps (iterating regime: Iv = 2 with warm start)"))
            # ... and the same regime at the headline's own size ("a second window ... violent phase"): the 16 M scene leaves
            # Id = Iv = 1 around step 1 900 and runs at Iv = 2 with a divergence warm start in every step from ~2 100 to ~3 300
            # (tools/iter_trace.py, profiles/r05_iter_trace_16M.txt); 2 500 untimed steps cost ~5 s
            if abs(head["n"] - 16_000_000) < 300_000:
                also.append(window_summary(single_window(args.particles, 2500, 20, 2),
                                           "DFSPH 16 M particles after 2500 steps (iterating regime at the headline's size: Iv = 2 with warm start)"))
        n_global, n, scale = head["n"], head["n"], head["scale"]
        out = {
            "metric": "particle-steps/sec (whole node), 2D DFSPH dam-break" if args.solver == "dfsph" else "particle-steps/sec, 2D WCSPH dam-break",
            "value": head["value"],
            "unit": "particle-steps/s",
            "n_gpus": 1,
            "world_size_seen": 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.solver.upper()} 2D dam-break (main.rs:177-196 scene x{scale:.2f}), {n_global} fluid + {head['n_boundary']} boundary particles "
                            f"on one GPU, adaptive CFL timer from t=0"
                            + (f" = {config_name[1]}" if (args.solver == "dfsph" and abs(n_global - 1_000_000) < 20_000 and not args.skip_steps) else "")
                            + (f" = {config_name[2]}" if (args.solver == "dfsph" and abs(n_global - 16_000_000) < 300_000 and not args.skip_steps) else "")
                            + (f", window after {args.skip_steps} steps" if args.skip_steps else "")
                            + (f", fixed iterations {tuple(args.fixed_iterations)}" if any(args.fixed_iterations) else "")
                            + (f", solver tolerances x{args.tolerance_scale}" if args.tolerance_scale != 1.0 else "") + ", two-phase step through the C ABI",
                "particles_per_gpu": n,
                "prewarm": f"a scratch context keeps the GPU busy from {args.prewarm_ms:.0f} ms before the first warm-up step until that step (the upload) has returned: no idle gap in front of the timed region (BusyGpu in bench.py)" if args.prewarm_ms > 0 else "none",
                "particles_total": n_global,
                "parallelism": "single GPU",
                "mean_density_iterations": head["it"]["Id"], "mean_divergence_iterations": head["it"]["Iv"], "warmstart_rate": [head["it"]["Wd"], head["it"]["Wv"]],
                "mean_neighbors": head["kbar"],
                "max_density_iterations_seen": head["it"]["Id_max"], "max_divergence_iterations_seen": head["it"]["Iv_max"],
                "host_calls": "one library call runs the K timed steps (sphx_*_simulation_steps: the caller's frame loop, main.rs:348-350, "
                              "in C; every step is a full Solver::simulation_step)" if not (args.per_step_calls or args.no_device_dt or args.abi_calls)
                else "one library call per step from Python",
                "solver_loop": "host-run (SPHX_HOST_LOOP=1)" if os.environ.get("SPHX_HOST_LOOP") == "1" else
                "device-run (residual test on the device, iterations queued ahead)",
            },
            "step_model": head["model"],
        }
        if head["roof"]:
            out["roofline"] = head["roof"]
        if also:
            out["also"] = also
        if not args.no_cpu_baseline and args.solver == "dfsph":
            # a bounded sample of the headline's OWN workload: the same scene at the same size, one warm-up step and two
            # timed ones per variant (~6 s per step of the reference-faithful variant at 16 M on the GPU box's host); beside it the 1 M
            # sample of rounds 1-4, so that the line shows what the size does to the restatement's cost per particle-step
            out["cpu_baseline"] = cpu_baseline(scale, budget_s=60.0, max_steps=2 if scale > float(np.sqrt(1_500_000 / 4050.0)) else 10)
            if scale > float(np.sqrt(1_500_000 / 4050.0)):
                small = cpu_baseline(float(np.sqrt(1_000_000 / 4050.0)), budget_s=4.0, max_steps=10)
                out["cpu_baseline"]["at_1M_particles"] = {k: small.get(k) for k in ("value", "unit", "cores", "sample")}
                out["cpu_baseline"]["at_1M_particles"]["all_parallel"] = (small.get("all_parallel") or {}).get("value")
                if small.get("value") and out["cpu_baseline"].get("value"):
                    out["cpu_baseline"]["cost_per_particle_step_vs_1M"] = small["value"] / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        busy.close()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
        return

    # ---- tiles: N > 1 ranks (one per GPU), or --force-tiles on one ----------------------------------------------------------------
    if args.solver == "wcsph":
        raise SystemExit("--solver wcsph runs on one GPU")
    # weak scaling: the global scene holds `particles` per GPU
    scale, w = scene_of(args.particles * world)
    pos, boundary = w.positions, w.boundary_particles
    t_scene_done = time.perf_counter()
    n_global = len(pos)
    diam = np.float32(2.0) * np.float32(w.properties()["particle_radius"])
    timer = y.TimeManager()
    params = make_params()
    # The tile step loop runs INSIDE libsphx (sphx_multi, csrc/sphx_tiles.cpp): this process holds one tile.  Transport of the halo
    # records: the library's own grouped ncclSend/ncclRecv (RCCL over xGMI) with shared-memory scalars, or — --comm torch, and
    # always over gloo — torch.distributed through the sphx_comm_ops function table.
    from yasph2d_amd import _lib as ylib
    from yasph2d_amd.multi import MultiSolver, TorchCommOps

    lay = {"auto": ylib.LAYOUT_AUTO, "strips": ylib.LAYOUT_STRIPS, "grid": ylib.LAYOUT_GRID}[args.tiles]
    kw = dict(halo=args.halo, fixed_halo=args.fixed_halo, rebalance_every=args.rebalance_every, layout=lay, overlap_exchange=args.overlap_exchange)
    job = "bench" + os.environ.get("MASTER_PORT", "0")
    if dist is not None:
        # a per-run token in the segment's name on top of the library's own stale-segment handshake
        tok = [os.urandom(4).hex() if rank == 0 else None]
        dist.broadcast_object_list(tok, src=0)
        job += "_" + tok[0]
    if dist is None:
        multi = MultiSolver(params, devices=[dev_index], **kw)  # --force-tiles: one tile, the tile code path
    else:
        use_builtin = args.comm == "rccl" or (args.comm == "auto" and args.backend == "nccl")
        if use_builtin:
            try:
                multi = MultiSolver.rank(params, dev_index, rank, world, comm=None, job=job, **kw)
                ok = 1.0
            except y.SphxError as e:
                # (the library's bring-up fails on EVERY rank together — status rounds over the shared segment — so nobody hangs here)
                sys.stderr.write(f"bench.py: rank {rank}: built-in RCCL transport failed ({e}); falling back to torch.distributed\n")
                ok = 0.0
            t = torch.tensor([ok], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)  # all ranks take the same decision
            if float(t.item()) < 1.0:
                if multi is not None:
                    multi.close()
                multi, use_builtin = None, False
        if not use_builtin:
            comm = TorchCommOps(dist, torch.device("cuda", dev_index), shm_name=job + "t" if args.scalar_comm == "shm" else None)
            multi = MultiSolver.rank(params, dev_index, rank, world, comm=comm, **kw)
    def diag(stage):
        # what a maintainer needs when an N > 1 run dies (nobody has seen one on real links yet): printed on EVERY failure path
        try:
            rv = ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else "?"
        except Exception as ex:  # noqa: BLE001
            rv = f"unavailable ({ex})"
        seen = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        tr = None
        try:
            tr = multi.info()["transport"] if multi is not None else None
        except Exception:  # noqa: BLE001
            pass
        sys.stderr.write(f"bench.py: rank {rank}/{world} failed in {stage}: backend={args.backend} comm={args.comm} transport={tr} world_size_seen={seen} "
                         f"RCCL={rv} device={dev_index} HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')} "
                         f"particles_per_gpu={args.particles} t+{time.perf_counter() - t_process_start:.1f}s\n")

    try:
        multi.set_boundary(boundary)
        multi.upload(pos)
    except Exception:
        diag("scene upload")
        raise
    ctx = multi.tile_context(0)
    n = n_global // world
    setup_seconds = time.perf_counter() - t_process_start  # process start -> tiles resident (imports, rendezvous, scene build of the GLOBAL scene on every rank, upload)

    def one_step():
        st = multi.step(timer, diam)
        st["neighbor_entries"] = 0
        return st

    def k_steps(k):
        return multi.steps(timer, k, diam)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    try:
        elapsed, stats, roof = measure(one_step, k_steps, ctx, barrier, args.steps, args.warmup, args.skip_steps, not args.no_roofline, n, reduce_max)
    except Exception:
        diag("the step loop")
        raise
    it = iteration_stats(stats)
    minfo = multi.info()
    # measured list statistics of the tiles' latest neighbour build, summed over the ranks; per-rank owned counts
    tot = [float(minfo["neighbor_entries"]), float(minfo["remote_entries"]), float(minfo["build_particles"])]
    owned = [int(minfo["owned_local"])]
    if dist is not None:
        t = torch.tensor(tot, dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t)
        tot = t.tolist()
        gathered = [None] * world
        dist.all_gather_object(gathered, owned[0])
        owned = [int(v) for v in gathered]
    kbar = tot[0] / tot[2] if tot[2] else None
    rbar = tot[1] / tot[2] if tot[2] else None
    if rank == 0:
        value = n_global * args.steps / elapsed
        out = {
            "metric": "particle-steps/sec (whole node), 2D DFSPH dam-break",
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "world_size_seen": dist.get_world_size() if dist is not None else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"DFSPH 2D dam-break (main.rs:177-196 scene x{scale:.2f}), {n_global} fluid + {len(boundary)} boundary particles "
                            f"in total ({n} fluid per GPU), adaptive CFL timer from t=0"
                            + (f" = {config_name[world]}" if (world in (4, 8) and abs(n - 16_000_000) < 300_000 and not args.skip_steps) else "")
                            + (f", window after {args.skip_steps} steps" if args.skip_steps else "")
                            + (f", fixed iterations {tuple(args.fixed_iterations)}" if any(args.fixed_iterations) else "")
                            + (f", solver tolerances x{args.tolerance_scale}" if args.tolerance_scale != 1.0 else "") + ", two-phase step through the C ABI",
                "particles_per_gpu": n,
                "owned_particles_per_rank": owned,
                "prewarm": f"a scratch context keeps the GPU busy from {args.prewarm_ms:.0f} ms before the first warm-up step until that step (the upload) has returned: no idle gap in front of the timed region (BusyGpu in bench.py)" if args.prewarm_ms > 0 else "none",
                "particles_total": n_global,
                "transport": minfo["transport"],
                "setup_seconds": setup_seconds,
                "setup_breakdown_seconds": {"imports_rendezvous_and_global_scene_build": t_scene_done - t_process_start,
                                            "solver_create_boundary_and_upload": setup_seconds - (t_scene_done - t_process_start),
                                            "of_which_ownership_pass_over_the_global_scene_in_libsphx": minfo.get("ownership_seconds")},
                # what one halo exchange moves out of this rank (all peers together): `packed` = (1 + records) * 32 B per peer as counted on
                # the device, `sent` = what ncclSend was given (packed rounded up to 64 KiB when the record counts travel first —
                # SPHX_EXACT_EXCHANGE, default for messages >= 1 MiB at capacity — else the buffers' capacity; packed is 0 then: not known)
                "halo_bytes_per_step_per_rank": {"packed": minfo["halo_bytes_packed"] / max(1, minfo["exchanges"]),
                                                 "sent": minfo["halo_bytes_sent"] / max(1, minfo["exchanges"]),
                                                 "capacity_per_peer": (1 + minfo["cap_records"]) * 32},
                "parallelism":
                f"{world} spatial tiles ({'columns cut again across (2 x N/2)' if minfo['grid_layout'] else 'strips along ' + 'xy'[max(minfo['axis'], 0)]}, cut at "
                f"particle-count quantiles), step loop inside libsphx (sphx_multi), ghost halo {minfo['halo_now']} of <= {args.halo} cells (follows "
                f"the ring budget), per step: 1 halo exchange with {minfo['peers']} neighbours + 3 scalar all-reduces; transport: "
                f"{minfo['transport']}; {minfo['exchanges']} exchanges ({minfo['band_packs']} of them packed from the density correction's "
                f"classification: only the send bands were visited again) and {minfo['rebalances']} re-partitions in total",
                "mean_density_iterations": it["Id"], "mean_divergence_iterations": it["Iv"], "warmstart_rate": [it["Wd"], it["Wv"]], "mean_neighbors": kbar,
                "max_density_iterations_seen": it["Id_max"], "max_divergence_iterations_seen": it["Iv_max"],
                "host_calls": "one library call runs the K timed steps (sphx_*_simulation_steps: the caller's frame loop, main.rs:348-350, "
                              "in C; every step is a full Solver::simulation_step)" if not args.per_step_calls else "one library call per step from Python",
                "solver_loop": "tile loop: the verdict of every solver iteration needs the all-reduce over the tiles",
                "tile_path_cost_at_world_1": "the tile code path itself, measured with one tile on one GPU (bench.py --force-tiles, profiles/r04_bench_{1M,16M}_forcetiles.json): "
                                             "within +-2 % of the single context at 16 M particles per GPU (the size of the multi-GPU configs), +8 % at 1 M (three host "
                                             "decisions per step that the single context takes on the device)",
            },
            "step_model": step_model(kbar if kbar is not None else 8.0, rbar if rbar is not None else 0.5, it, n, args.steps, elapsed, kbar is not None),
        }
        if roof:
            out["roofline"] = roof
        sys.stdout.flush()
        busy.close()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    # tiles, their communicator and the shared-memory segment of the scalar all-reduce (rank 0 unlinks it)
    comm_obj = getattr(multi, "_comm", None)
    if dist is not None:
        dist.barrier()
    multi.close()
    if comm_obj is not None and hasattr(comm_obj, "close"):
        comm_obj.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
