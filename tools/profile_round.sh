#!/bin/bash
# Runs on the GPU box (gpurun): the rocprofv3 passes whose summaries are committed under profiles/.
#   tools/profile_round.sh OUTDIR
# Every pass is the same command the bench contract names (python3 bench.py ...), one counter group per pass, each under its own timeout.
out=$GRAFT_REPO_ROOT/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline"
run() { name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d $out/$name -- $B $ARGS > $out/$name.log 2>&1; echo "$name rc=$?"; }
ARGS="--steps 100 --warmup 5"
run stats_1M --kernel-trace --stats
ARGS="--steps 20 --warmup 2"
run fetch_1M --pmc FETCH_SIZE --kernel-trace
run write_1M --pmc WRITE_SIZE --kernel-trace
run sq_1M --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace
run tcc_1M --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace
ARGS="--steps 20 --warmup 2 --particles 16000000"
run stats_16M --kernel-trace --stats
run fetch_16M --pmc FETCH_SIZE --kernel-trace
run write_16M --pmc WRITE_SIZE --kernel-trace
cd $GRAFT_REPO_ROOT
S=tools/summarize_profile.py
for n in stats_1M stats_16M; do f=$(find $out/$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $S $f > $out/$n.txt; done
for n in fetch_1M write_1M fetch_16M write_16M; do f=$(find $out/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $S $f > $out/$n.txt; done
for n in sq_1M tcc_1M; do f=$(find $out/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 tools/pmc_table.py $f > $out/$n.txt; done
f=$(find $out/stats_1M -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/gaps.py $f > $out/gaps_1M.txt
find $out -name "*.csv" -size +1M -delete
ls $out
