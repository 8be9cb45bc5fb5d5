#!/bin/bash
# forced-tile path on one GPU: classification fused into the last density correction (default) against SPHX_TILE_FUSE_CLASS=0, and the single context
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  ms/step %.4f' % d['ms_per_step']); print('  ', {k[:24]:round(v*1000,1) for k,v in d['roofline'].get('per_kernel_ms_per_step_event_inflated',{}).items()})"; }
for size in "--steps 200" "--particles 16000000 --steps 20 --warmup 3"; do
  echo "== $size"
  echo " single"; run $size
  echo " tiles fused"; run $size --force-tiles
  echo " tiles unfused"; SPHX_TILE_FUSE_CLASS=0 run $size --force-tiles
done
