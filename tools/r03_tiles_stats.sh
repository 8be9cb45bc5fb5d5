#!/bin/bash
# rocprofv3 kernel statistics of the forced-tile path (one tile) at 1 M and 16 M
out=$GRAFT_REPO_ROOT/gpurun_out/tiles_stats; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0 --force-tiles"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t1 -- $B --steps 100 --warmup 5 > $out/t1.log 2>&1; echo rc=$?
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t16 -- $B --steps 20 --warmup 2 --particles 16000000 > $out/t16.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
for n in t1 t16; do f=$(find $out/$n -name "*kernel_stats.csv" | head -1); python3 tools/summarize_profile.py $f > $out/stats_$n.txt; done
find $out -name "*.csv" -delete
head -14 $out/stats_t1.txt; head -14 $out/stats_t16.txt
