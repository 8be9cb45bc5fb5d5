#!/bin/bash
# tools/r02_ab.sh OUT VARIANT...  — bench lines (fixed iterations 1 1: timing only) of the product library and the named variants
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$PWD/yasph2d_amd/variants/libsphx_$v.so; fi
  for sz in 1000000 16000000; do
    st=200; [ $sz = 16000000 ] && st=20
    timeout 300 python bench.py --steps $st --warmup 3 --no-cpu-baseline --fixed-iterations 1 1 --particles $sz > $out/${v}_$sz.json 2> $out/${v}_$sz.err
    python3 - $out/${v}_$sz.json $v $sz <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], sys.argv[3], 'G=%.3f'%(d['value']/1e9), 'ms=%.4f'%d['ms_per_step'], {k.replace('correct_velocity_with_','CO_').replace('compute_','CE_')[:18]:round(v*1000,1) for k,v in d['roofline']['per_kernel_ms_per_step_event_inflated'].items()})
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
  done
done
