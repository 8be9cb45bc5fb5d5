#!/bin/bash
# Same-box A/B of library variants (tools/ab_build.sh NAME ...):  tools/ab_variants.sh [-r ROUNDS] default NAME1 NAME2 ...
# "default" is the product library.  Rounds are interleaved (boxes drift by a per cent within minutes); one discarded run first.
cd $GRAFT_REPO_ROOT
rounds=3
if [ "$1" = "-r" ]; then rounds=$2; shift 2; fi
run() {  # variant label args
  local lib=""; [ "$1" != default ] && lib=$PWD/yasph2d_amd/variants/libsphx_$1.so
  SPHX_LIB=$lib python3 bench.py --no-cpu-baseline --no-roofline --no-also $3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s %-9s' % ('$1', '$2'), round(d['ms_per_step']*1000,1), 'us/step', round(d['value']/1e9,3), 'G/s')"
}
run default discard "--steps 30 --warmup 5" >/dev/null
for r in $(seq $rounds); do
  for v in "$@"; do run $v "16M" "--steps 100 --warmup 10"; done
done
for r in $(seq $rounds); do
  for v in "$@"; do run $v "1M" "--steps 300 --warmup 30 --particles 1000000"; done
done
for r in 1 2; do
  for v in "$@"; do run $v "16M-late" "--steps 20 --warmup 2 --skip-steps 2500"; done
done
