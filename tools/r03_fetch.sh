#!/bin/bash
# HBM traffic per launch (FETCH_SIZE / WRITE_SIZE passes) of the product library and variants:  BENCH_ARGS="--particles 16000000" tools/r03_fetch.sh OUT VARIANT...
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$GRAFT_REPO_ROOT/yasph2d_amd/variants/libsphx_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${c}_$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0 --steps 10 --warmup 2 $BENCH_ARGS > $out/${c}_$v.log 2>&1; echo "$v $c rc=$?"
    f=$(find $out/${c}_$v -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $f > $out/${c}_$v.txt
    find $out/${c}_$v -name "*.csv" -delete
  done
  echo "== $v"; head -12 $out/FETCH_SIZE_$v.txt
done
