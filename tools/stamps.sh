#!/bin/bash
# GPU box: in-kernel phase stamps of the neighbour build for variants built with -DSPHX_STAMPS.  tools/stamps.sh OUTNAME VARIANT... [-- PARTICLES...]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $out; cd $GRAFT_REPO_ROOT
vs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vs+=("$1"); shift; done; [ "$1" = "--" ] && shift
ps=("$@"); [ ${#ps[@]} -eq 0 ] && ps=(1000000 16000000)
for v in "${vs[@]}"; do
  export SPHX_LIB=$GRAFT_REPO_ROOT/yasph2d_amd/variants/libsphx_$v.so
  for P in "${ps[@]}"; do
    timeout 300 python3 bench.py --steps 20 --warmup 2 --particles $P --no-cpu-baseline --no-also --no-roofline --prewarm-ms 0 ${STAMP_ARGS} > $out/bench_${v}_$P.json 2> $out/bench_${v}_$P.err
    echo "$v $P: $(grep SPHX_STAMPS $out/bench_${v}_$P.err | tail -4 | tr "\n" " ")"
  done
done
