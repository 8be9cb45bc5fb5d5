#!/bin/bash
# kernel trace of the world-1 tile path next to the single-context path: per-kernel stats and idle gaps
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-single tiles}; do
  extra=""; [ $mode = tiles ] && extra="--force-tiles"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$mode -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --steps 100 --warmup 5 --prewarm-ms 0 $extra "$@" > $out/$mode.log 2>&1
  f=$(find $out/$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $f > $out/$mode.stats.txt
  f=$(find $out/$mode -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/gaps.py $f > $out/$mode.gaps.txt
  grep '^{' $out/$mode.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$mode', d['ms_per_step'])"
  cat $out/$mode.stats.txt | head -30; cat $out/$mode.gaps.txt | head -40
done
find $out -name "*.csv" -size +1M -delete
