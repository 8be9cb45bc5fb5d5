#!/bin/bash
# GPU box: the default bench line end to end (what the driver runs), with its wall time.  tools/r05_default.sh OUTNAME
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
t0=$(date +%s)
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
python3 - $out/bench_default.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        r = d["roofline"]
        print(round(d["value"] / 1e9, 3), "G/s", d["ms_per_step"], "ms; bound", r["bound"], "frac", round(r["frac"], 4), "avg_launch_ms", round(r["avg_launch_ms"], 4), "traffic", r.get("traffic"), "valu_issue_frac", r.get("valu_issue_frac"))
        for a in d.get("also", []): print("  also:", a["window"][:70], round(a["value"] / 1e9, 3), "G/s Iv", a["mean_divergence_iterations"], "dominant", (a.get("dominant_kernel") or {}).get("kernel"))
        c = d["cpu_baseline"]
        print("  cpu:", {k: c.get(k) for k in ("value", "cores", "sample", "cost_per_particle_step_vs_1M")}); print("  cpu 1M:", c.get("at_1M_particles")); print("  cpu all_parallel:", c["all_parallel"]["value"])
PY
