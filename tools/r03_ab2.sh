#!/bin/bash
# alternating A/B of the product library against variants: N rounds of (base, variant...) at 1 M from t = 0, in the late window and at 16 M
cd $GRAFT_REPO_ROOT
rounds=${ROUNDS:-3}
run() { python bench.py --no-cpu-baseline --no-also --no-roofline $EXTRA "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4f' % d['ms_per_step'], end=' ')"; }
for w in "--steps 300" "--steps 200 --skip-steps 3750" "--steps 30 --warmup 3 --particles 16000000"; do
  echo "== $w"
  for v in base "$@"; do
    if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$PWD/yasph2d_amd/variants/libsphx_$v.so; fi
    echo -n "$v: "; for r in $(seq $rounds); do run $w; done; echo
  done
  for v in base "$@"; do
    if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$PWD/yasph2d_amd/variants/libsphx_$v.so; fi
    echo -n "$v: "; for r in $(seq $rounds); do run $w; done; echo
  done
done
