#!/bin/bash
# the tile path (one tile, --force-tiles) and the single context through the late window (3750 steps, then 200 timed)
cd $GRAFT_REPO_ROOT
for a in "--force-tiles" ""; do
  python bench.py --no-cpu-baseline --no-also --no-roofline --steps 200 --skip-steps 3750 $a 2>gpurun_out/tl.err | python -c "
import sys,json
l=[x for x in sys.stdin if x.startswith('{')]
if not l: print('$a', 'NO LINE'); sys.exit()
d=json.loads(l[-1]); print('$a', d['ms_per_step'], d['value']/1e9, d['config'].get('mean_divergence_iterations'))"
  tail -2 gpurun_out/tl.err
done
