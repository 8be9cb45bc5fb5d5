#!/bin/bash
# GPU box: the late (iterating) window at 1 M + the CPU baseline's per-phase seconds.  tools/r04_late.sh OUTNAME
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 200 --particles 1000000 --skip-steps 3750 --no-cpu-baseline --no-also > $out/bench_late.json 2> $out/bench_late.err; echo "late rc=$?"
python3 - $out/bench_late.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e9,3), round(d['ms_per_step'],4), {k[:34]:round(v*1000,1) for k,v in d['roofline']['per_kernel_ms_per_step_event_inflated'].items()})
PY
python3 - > $out/cpu_phases.txt 2>&1 <<'PY'
import sys, json
sys.path.insert(0, '.')
import bench, numpy as np
r = bench.cpu_baseline(float(np.sqrt(1_000_000 / 4050.0)))
print(json.dumps(r, indent=1))
PY
head -60 $out/cpu_phases.txt
