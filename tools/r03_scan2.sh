#!/bin/bash
# one-launch (look-back) scan against the two-launch scan (SPHX_SCAN_TWO_PASS=1): cell_scan us/launch and ms/step
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); pk=d['roofline']['per_kernel_ms_per_step_event_inflated']
print('  ms/step %.4f  cell_scan %.1f us' % (d['ms_per_step'], pk['cell_scan']*1000))"; }
for w in "--steps 200" "--steps 200 --skip-steps 3750" "--steps 20 --warmup 3 --particles 16000000"; do
  echo "== $w"; echo " one launch"; run $w; echo " two launches"; SPHX_SCAN_TWO_PASS=1 run $w
done
