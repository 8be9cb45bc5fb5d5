#!/bin/bash
# GPU box: alternating plain bench lines of library variants for given bench arguments.
#   tools/r05_ab_bench.sh OUTNAME "BENCH ARGS" ROUNDS VARIANT...     ("base" = the product library)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; args=$2; rounds=$3; shift 3
mkdir -p $out; cd $GRAFT_REPO_ROOT
for round in $(seq 1 $rounds); do for v in "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$GRAFT_REPO_ROOT/yasph2d_amd/variants/libsphx_$v.so; fi
  timeout 600 python3 bench.py $args --no-cpu-baseline --no-also > $out/bench_${v}_$round.json 2> $out/bench_${v}_$round.err
  python3 - $out/bench_${v}_$round.json $v $round <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{}); pk=r.get('per_kernel_ms_per_step_event_inflated',{})
        print(sys.argv[2], "round", sys.argv[3], round(d['value']/1e9,3), "G/s", round(d['ms_per_step'],4), "ms", "Id/Iv", d['config']['mean_density_iterations'], d['config']['mean_divergence_iterations'], {k[:22]:round(v*1000,1) for k,v in pk.items()})
PY
done; done
