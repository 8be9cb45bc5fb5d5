#!/bin/bash
# SQ counters (instructions per wavefront) of the product library and of variants:  tools/r03_sq.sh OUTNAME [bench args --] VARIANT...
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$GRAFT_REPO_ROOT/yasph2d_amd/variants/libsphx_$v.so; fi
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/sq_$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --prewarm-ms 0 --steps 20 --warmup 2 $BENCH_ARGS > $out/sq_$v.log 2>&1; echo "$v rc=$?"
  f=$(find $out/sq_$v -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $f > $out/sq_$v.txt
  find $out/sq_$v -name "*.csv" -delete
  echo "== $v"; head -8 $out/sq_$v.txt
done
