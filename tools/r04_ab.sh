#!/bin/bash
# GPU box: same-box A/B of library variants (tools/ab_build.sh) at 16 M and 1 M.  tools/r04_ab.sh OUTNAME VARIANT...
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $out; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$PWD/yasph2d_amd/variants/libsphx_$v.so; fi
  for P in 16000000 1000000; do
  timeout 300 python bench.py --steps $([ $P = 1000000 ] && echo 200 || echo 40) --particles $P --no-cpu-baseline --no-also > $out/b_${v}_${P}_$rep.json 2> $out/b_${v}_${P}_$rep.err
  python3 - $out/b_${v}_${P}_$rep.json $v $P <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2].ljust(8), sys.argv[3].ljust(9), round(d['value']/1e9,3), round(d['ms_per_step'],4), {k[:18]:round(v*1000,1) for k,v in d['roofline']['per_kernel_ms_per_step_event_inflated'].items()})
PY
  done
done
done
