#!/bin/bash
cd $GRAFT_REPO_ROOT
export SPHX_LIB=$PWD/yasph2d_amd/variants/libsphx_stamps.so
for sz in 1000000 16000000; do
  st=100; [ $sz = 16000000 ] && st=10
  timeout 300 python bench.py --steps $st --warmup 2 --no-cpu-baseline --no-roofline --particles $sz 2>&1 >/dev/null | grep SPHX_STAMPS
done
timeout 300 python bench.py --steps 100 --warmup 2 --skip-steps 3750 --no-cpu-baseline --no-roofline 2>&1 >/dev/null | grep SPHX_STAMPS
