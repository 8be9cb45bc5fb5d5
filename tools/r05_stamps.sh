#!/bin/bash
# GPU box: in-kernel phase stamps of the neighbour build (variant "stamps" = -DSPHX_STAMPS).  tools/r05_stamps.sh OUTNAME
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
export SPHX_LIB=$GRAFT_REPO_ROOT/yasph2d_amd/variants/libsphx_stamps.so
for P in 1000000 16000000; do
  timeout 300 python3 bench.py --steps 20 --warmup 2 --particles $P --no-cpu-baseline --no-also --no-roofline --prewarm-ms 0 > $out/bench_$P.json 2> $out/bench_$P.err
  grep SPHX_STAMPS $out/bench_$P.err | tail -3
done
