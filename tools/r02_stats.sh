#!/bin/bash
# kernel-trace stats + inter-kernel gaps of one bench command:  tools/r02_stats.sh OUT [bench args...]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 5 "$@" > $out/stats.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 tools/summarize_profile.py $f > $out/stats.txt
f=$(find $out/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/gaps.py $f > $out/gaps.txt
find $out -name "*.csv" -size +1M -delete
cat $out/stats.txt | head -24; cat $out/gaps.txt | tail -25
