#!/bin/bash
# tools/ab_run.sh OUTDIR VARIANT...   — bench.py at 1M for the product library ("base") and each named variant
out=$1; shift
mkdir -p $out
for v in base "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$PWD/yasph2d_amd/variants/libsphx_$v.so; fi
  timeout 300 python bench.py --steps 100 --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err
  python3 - $out/bench_$v.json $v <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], round(d['value']/1e9,3), round(d['ms_per_step'],4), {k[:14]:round(v*1000,1) for k,v in d['roofline']['per_kernel_ms_per_step_event_inflated'].items()})
PY
done
