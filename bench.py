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
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0  # what a float4 copy reaches on it (same guide): the practical ceiling of a streaming kernel


def bytes_per_particle_step(kbar, Id, Iv, Wd, Wv, compressed=False, rbar=0.5):
    """SURVEY.md §8(d) / BASELINE.md §4 list-based algorithmic bytes per particle-step.  compressed: the workgroup-local 16-bit
    list format of this build — every list-consuming traversal moves the count word (4), 2*kbar of entries and 4*rbar of out-of-window
    table lines (rbar = such entries per particle, ~0.5) instead of the 8 + 4*kbar of the 32-bit lists."""
    if compressed:
        save = (8 + 4 * kbar) - (4 + 2 * kbar + 4 * rbar + 4.0 / 256)  # per traversal
        return 252 + 16 * kbar - 4 * save + Id * (84 + 8 * kbar - 2 * save) + Iv * (80 + 8 * kbar - 2 * save) + (Wd + Wv) * (44 + 4 * kbar - save)
    return 252 + 16 * kbar + Id * (84 + 8 * kbar) + Iv * (80 + 8 * kbar) + (Wd + Wv) * (44 + 4 * kbar)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pos, boundary, budget_s=12.0, max_steps=20):
    """The oracle's OpenMP build (the reference's Rayon loops restated; 'port') timed on this host's cores, threads pinned
    (OMP_PROC_BIND=close, OMP_PLACES=cores: without pinning the figure swung 1.9-3.5 M particle-steps/s between runs in round 1)."""
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    from oracle.oracle import Oracle, lib

    L = lib(omp=True)
    cores = L.orc_get_max_threads()
    o = Oracle(omp=True)
    o.set_boundary(boundary)
    o.set_particles(pos)
    o.dfsph_step()  # includes the warm-up block, like the GPU warm-up steps
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps:
        o.dfsph_step()
        steps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return {
        "value": len(pos) * steps / el,
        "unit": "particle-steps/s",
        "cores": cores,
        "kind": "port",
        "cpu": cpu_model(),
        "threads": "OpenMP, OMP_PROC_BIND=%s OMP_PLACES=%s" % (os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES")),
        "sample": f"{steps} DFSPH steps of the same {len(pos)}-particle dam-break after 1 warm-up step, "
                  f"C++/OpenMP restatement of yasph2d's Rayon path (not the Rust binary), {el:.1f} s",
    }


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
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--skip-steps", type=int, default=0,
                    help="untimed steps before the warm-up (SURVEY 8(d): a second window after 2000 steps = the violent phase, high Id/Iv)")
    ap.add_argument("--particles", type=int, default=1_000_000, help="target fluid particles per GPU (configs[1] = 1M)")
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
    ap.add_argument("--lists-32bit", action="store_true", help="disable the 16-bit neighbour-list compression (A/B runs)")
    ap.add_argument("--solver", default="dfsph", choices=["dfsph", "wcsph"],
                    help="wcsph: the second Solver of the reference (solver/wscsph.rs, cfl factor 0.2, main.rs:116-119) on one GPU")
    ap.add_argument("--no-device-dt", action="store_true", help="plain sphx_step_begin: the device waits for the host's dt (A/B runs)")
    ap.add_argument("--fixed-iterations", type=int, nargs=2, default=(0, 0), metavar=("ID", "IV"),
                    help="run exactly ID constant-density and IV divergence iterations per step (0 0 = adaptive, the reference's behaviour)")
    ap.add_argument("--tolerance-scale", type=float, default=1.0,
                    help="multiply both solver tolerances (dfsph.rs:49,53) by this; < 1 makes the loops iterate (the iterating-regime window)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="run a scratch context for this long before the measured one is created (GPU clocks / first-touch; 0 = off)")
    ap.add_argument("--per-step-calls", action="store_true", help="timed region: one library call per step from Python instead of one call "
                    "that runs the K steps (sphx_solver_simulation_steps: the caller's frame loop, main.rs:348-350, inside the library)")
    ap.add_argument("--abi-calls", action="store_true", help="drive the two-phase C ABI from Python (one ctypes call per phase) instead of "
                                                             "the host mirror's one call per step (A/B runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher self-test: ranks rendezvous (gloo), barrier, all-reduce and print the line with value = null; no GPU, no compute")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
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
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seen = torch.tensor([1.0], dtype=torch.float64)
            dist.all_reduce(seen)  # every rank arrived
            assert int(seen.item()) == world
        if rank == 0:
            os.write(result_fd, (json.dumps({"metric": "particle-steps/sec (whole node), 2D DFSPH dam-break", "value": None,
                                             "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                             "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                                             "dtype": "f32", "data": "none (dry run of the rank launcher)",
                                             "config": {"workload": "dry run: no compute"}}) + "\n").encode())
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

    # weak scaling: the global scene holds `particles` per GPU
    scale = float(np.sqrt(args.particles * world / 4050.0))
    w = y.FluidParticleWorld()
    w.reset_fluid(scale)
    pos, boundary = w.positions, w.boundary_particles
    n_global = len(pos)
    diam = np.float32(2.0) * np.float32(w.properties()["particle_radius"])
    timer = y.TimeManager(cfl_factor=0.2) if args.solver == "wcsph" else y.TimeManager()
    params = y.default_params(device=dev_index, fixed_iterations=tuple(args.fixed_iterations))
    if args.tolerance_scale != 1.0:
        params.max_avg_density_error = float(np.float32(params.max_avg_density_error) * np.float32(args.tolerance_scale))
        params.max_divergence_error = float(np.float32(params.max_divergence_error) * np.float32(args.tolerance_scale))
    if args.lists_32bit:
        params.list_span_limit = y.LISTS_32BIT
    if args.solver == "wcsph" and (world > 1 or args.force_tiles):
        raise SystemExit("--solver wcsph runs on one GPU")
    if args.prewarm_ms > 0:
        # Clock / page warm-up on a SCRATCH context (its own small scene, destroyed before the measured one exists): a fresh box runs
        # its first few hundred milliseconds of kernels at lower clocks, and W = 5 warm-up steps are 1 ms.  The measured context still
        # does exactly --warmup untimed and --steps timed steps from t = 0.
        sw = y.FluidParticleWorld()
        sw.reset_fluid(float(np.sqrt(min(args.particles, 1_000_000) / 4050.0)))
        sctx = y.SphxContext(y.default_params(device=dev_index))
        sctx.set_boundary(sw.boundary_particles)
        sctx.upload(sw.positions)
        stimer = y.TimeManager()
        t_end = time.perf_counter() + args.prewarm_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(20):
                v = sctx.step_begin(stimer.simulation_step(), stimer.law(diam))
                sctx.step_finish(y.duration_as_secs_f32(stimer.update_simulation_step(diam, v)))
        sctx.synchronize()
        sctx.close()
        del sw, sctx, stimer
    multi = None
    k_steps = None  # the K timed steps in ONE library call (the caller's frame loop on the library's side), where the path has it
    if world == 1 and not args.force_tiles:
        n = n_global
        if args.no_device_dt or args.abi_calls:
            # the raw two-phase C ABI, one ctypes call per phase (A/B runs)
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

    else:
        # The tile step loop runs INSIDE libsphx (sphx_multi, csrc/sphx_tiles.cpp): this process holds one tile.  Transport of the halo
        # records: the library's own grouped ncclSend/ncclRecv (RCCL over xGMI) with shared-memory scalars, or — --comm torch, and
        # always over gloo — torch.distributed through the sphx_comm_ops function table.
        from yasph2d_amd import _lib as ylib
        from yasph2d_amd.multi import MultiSolver, TorchCommOps

        lay = {"auto": ylib.LAYOUT_AUTO, "strips": ylib.LAYOUT_STRIPS, "grid": ylib.LAYOUT_GRID}[args.tiles]
        kw = dict(halo=args.halo, fixed_halo=args.fixed_halo, rebalance_every=args.rebalance_every, layout=lay, overlap_exchange=args.overlap_exchange)
        job = "bench" + os.environ.get("MASTER_PORT", "0")
        if dist is not None:
            # the shared-memory segment of the scalar all-reduce is found by name: a token chosen by rank 0 keeps a segment a crashed
            # earlier run with the same port may have left behind from being opened by a rank that gets there before rank 0
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
        multi.set_boundary(boundary)
        multi.upload(pos)
        ctx = multi.tile_context(0)
        n = n_global // world

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

    # Warm-up: every launch bracketed by hipEvents on the context's stream -> which kernel dominates a step.  (All ranks take the
    # same decision: the kernel mix is the same on every tile.)
    dominant = None
    wprof = None
    if not args.no_roofline and args.warmup > 0:
        ctx.profile_reset()
        ctx.profile_filter(None)
        ctx.profile_enable(True)
    for _ in range(args.skip_steps + args.warmup):
        one_step()
    if not args.no_roofline and args.warmup > 0:
        ctx.profile_enable(False)
        wprof = ctx.profile_get()
        dominant = max(wprof.items(), key=lambda kv: kv[1]["total_ms"])[0] if wprof else None
    if not args.no_roofline and dominant is None:
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
        stats = k_steps(args.steps)
    else:
        for _ in range(args.steps):
            stats.append(one_step())
    barrier()
    elapsed = time.perf_counter() - t0
    live = None
    if dominant is not None:
        ctx.profile_enable(False)
        live = ctx.profile_get().get(dominant)
        ctx.profile_filter(None)
        live_from = "timed region"
        if (live is None or not live["launches"]) and wprof and dominant in wprof:
            # fewer than 4 timed launches of the dominant kernel (--steps < 4): the event-bracketed warm-up steps stand in
            live, live_from = wprof[dominant], "warm-up steps (the timed region was too short to bracket a launch)"
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    Id = float(np.mean([s["density_iterations"] for s in stats]))
    Iv = float(np.mean([s["divergence_iterations"] for s in stats]))
    Wd = float(np.mean([s["warmstart_density"] for s in stats]))
    Wv = float(np.mean([s["warmstart_divergence"] for s in stats]))
    kbar = float(np.mean([s["neighbor_entries"] for s in stats])) / n if multi is None else None
    minfo = multi.info() if multi is not None else None

    roof = None
    if live is not None and live["launches"]:
        name, rec = dominant, live
        ev_ms = ctx.profile_event_overhead()  # what an EMPTY hipEvent bracket measures on this stream, same process
        avg_raw = rec["total_ms"] / rec["launches"]
        avg_ms = max(avg_raw - ev_ms, 1e-6)
        ach = rec["bytes"] / rec["launches"] / (avg_ms * 1e-3) / 1e9
        # HBM traffic of that kernel from the PMC counters: cannot be collected from inside the process; taken from the committed
        # rocprofv3 passes of this same command (profiles/) when the workload matches, else null.
        traffic, traffic_src = None, None
        import glob

        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_traffic_*.json"))):
            try:
                tj = json.load(open(tf))
                if abs(tj["workload_particles"] - n) < 0.02 * n and name in tj["bytes_per_launch"]:
                    traffic, traffic_src = tj["bytes_per_launch"][name]["total"], tj["source"]
            except (OSError, KeyError, ValueError):
                pass
        # per-kernel table (information only): a short extra pass with every launch timed, outside the timed region
        ctx.profile_reset()
        ctx.profile_enable(True)
        extra = max(10, min(args.steps, 30))
        for _ in range(extra):
            one_step()
        ctx.profile_enable(False)
        prof = ctx.profile_get()
        roof = {
            "bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "frac_of_achievable_6300": ach / HBM_ACHIEVABLE_GBS,
            "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": avg_ms, "avg_launch_ms_with_event_bracket": avg_raw,
            "event_bracket_overhead_ms": ev_ms, "launches": rec["launches"],
            "algorithmic_bytes_per_launch": rec["bytes"] / rec["launches"],
            "measured": "hipEvents around every 4th launch of this kernel inside the timed region, on the context's stream, minus the "
                        "elapsed time of an empty event bracket measured in the same process" if live_from == "timed region" else
                        "hipEvents around this kernel's launches, minus the elapsed time of an empty event bracket; taken from the " + live_from,
            "per_kernel_ms_per_step_event_inflated": {k: v["total_ms"] / extra for k, v in sorted(prof.items())},
            "note": "per_kernel_ms_per_step_event_inflated brackets EVERY launch with events (each adds the overhead above and keeps "
                    "kernels from overlapping their neighbours' tails): its sum exceeds ms_per_step; information only",
        }

    if rank == 0:
        value = n_global * args.steps / elapsed
        kb = kbar if kbar is not None else 8.0  # tiles do not report k; 8.0 = lattice value
        rb = float(np.mean([s.get("remote_entries", 0) for s in stats])) / n if multi is None else 0.5
        bstep_ref = bytes_per_particle_step(kb, Id, Iv, Wd, Wv)
        bstep = bytes_per_particle_step(kb, Id, Iv, Wd, Wv, compressed=not args.lists_32bit, rbar=rb)
        out = {
            "metric": "particle-steps/sec (whole node), 2D DFSPH dam-break" if args.solver == "dfsph" else "particle-steps/sec, 2D WCSPH dam-break",
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
                "workload": f"{args.solver.upper()} 2D dam-break (main.rs:177-196 scene x{scale:.2f}), {n_global} fluid + {len(boundary)} boundary particles "
                            f"in total ({n} fluid per GPU), adaptive CFL timer from t=0"
                            + (f", window after {args.skip_steps} steps" if args.skip_steps else "")
                            + (f", fixed iterations {tuple(args.fixed_iterations)}" if any(args.fixed_iterations) else "")
                            + (f", solver tolerances x{args.tolerance_scale}" if args.tolerance_scale != 1.0 else "") + ", two-phase step through the C ABI",
                "particles_per_gpu": n,
                "prewarm": f"{args.prewarm_ms:.0f} ms of steps on a scratch context before the measured context was created" if args.prewarm_ms > 0 else "none",
                "particles_total": n_global,
                "parallelism": "single GPU" if multi is None else
                f"{world} spatial tiles ({'columns cut again across (2 x N/2)' if minfo['grid_layout'] else 'strips along ' + 'xy'[max(minfo['axis'], 0)]}, cut at "
                f"particle-count quantiles), step loop inside libsphx (sphx_multi), ghost halo {minfo['halo_now']} of <= {args.halo} cells (follows "
                f"the ring budget), per step: 1 halo exchange with {minfo['peers']} neighbours + 3 scalar all-reduces; transport: "
                f"{minfo['transport']}; {minfo['exchanges']} exchanges and {minfo['rebalances']} re-partitions in total",
                "mean_density_iterations": Id, "mean_divergence_iterations": Iv, "warmstart_rate": [Wd, Wv], "mean_neighbors": kbar,
                "max_density_iterations_seen": int(max(s["density_iterations"] for s in stats)),
                "max_divergence_iterations_seen": int(max(s["divergence_iterations"] for s in stats)),
                "host_calls": "one library call runs the K timed steps (sphx_*_simulation_steps: the caller's frame loop, main.rs:348-350, "
                              "in C; every step is a full Solver::simulation_step)" if (k_steps is not None and not args.per_step_calls)
                else "one library call per step from Python",
                "solver_loop": "host-run (SPHX_HOST_LOOP=1)" if os.environ.get("SPHX_HOST_LOOP") == "1" else
                "device-run (residual test on the device, iterations queued ahead)" if multi is None else "host-run with an all-reduce per iteration",
            },
            "step_model": {"bytes_per_particle_step": bstep, "bytes_per_particle_step_32bit_lists": bstep_ref,
                           "list_format": "32-bit" if args.lists_32bit else "workgroup-local 16-bit slots (32-bit fallback per wavefront)",
                           "out_of_window_entries_per_particle": rb,
                           "achieved_GBs_whole_step_per_gpu": bstep * n * args.steps / elapsed / 1e9,
                           "frac_of_hbm_peak_whole_step": bstep * n * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                           "frac_of_achievable_6300_whole_step": bstep * n * args.steps / elapsed / 1e9 / HBM_ACHIEVABLE_GBS,
                           "note": "at 1 M particles the working set (~170 MB) sits inside the 256 MB Infinity Cache: the HBM roofline is "
                                   "a soft bound there; 16 M (--particles 16000000) is the size where it binds"},
        }
        if roof:
            out["roofline"] = roof
        if not args.no_cpu_baseline and world == 1 and args.solver == "dfsph":
            out["cpu_baseline"] = cpu_baseline(pos, boundary)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if multi is not None:  # tiles, their communicator and the shared-memory segment of the scalar all-reduce (rank 0 unlinks it)
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
