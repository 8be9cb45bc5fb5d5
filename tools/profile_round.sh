#!/bin/bash
# Runs on the GPU box (gpurun): the rocprofv3 passes and bench lines whose summaries are committed under profiles/ as <PREFIX>_*.
#   tools/profile_round.sh PREFIX OUTDIR [prof|bench]   e.g. tools/profile_round.sh r06 gpurun_out/r06_final prof; copy the records; ... bench
P=$1
out=$GRAFT_REPO_ROOT/$2
mkdir -p $out
MODE=$3   # prof: counter / trace passes only; bench: bench lines only (run it AFTER the passes' JSON records have been copied to profiles/: bench.py quotes them); empty: both
cd /tmp && export TMPDIR=/tmp
# (--prewarm-ms 0: the profile holds the measured context's launches only, not the scratch context's clock warm-up)
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0"
run() { [ "$MODE" = bench ] && return; name=$1; shift; timeout 900 rocprofv3 "$@" --output-format csv -d $out/$name -- $B $ARGS > $out/$name.log 2>&1; echo "$name rc=$?"; }
SQ="--pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"
ARGS="--steps 100 --warmup 5 --particles 1000000"
run stats_1M --kernel-trace --stats
ARGS="--steps 20 --warmup 2 --particles 1000000"
run fetch_1M --pmc FETCH_SIZE --kernel-trace
run write_1M --pmc WRITE_SIZE --kernel-trace
run sq_1M $SQ --kernel-trace
ARGS="--steps 20 --warmup 2"   # the default workload: 16 M (BASELINE configs[2])
run stats_16M --kernel-trace --stats
run fetch_16M --pmc FETCH_SIZE --kernel-trace
run write_16M --pmc WRITE_SIZE --kernel-trace
run sq_16M $SQ --kernel-trace
# the iterating regime: fixed 3 + 2 iterations per step (warm starts on), and the adaptive window where Iv = 2
ARGS="--steps 100 --warmup 5 --particles 1000000 --fixed-iterations 3 2"
run stats_1M_iterating_fixed32 --kernel-trace --stats
ARGS="--steps 100 --warmup 5 --particles 1000000 --skip-steps 3750"
run stats_1M_iterating_window3750 --kernel-trace --stats
# the same regime at the headline's size: Iv = 2 with a divergence warm start in every step; counters of the LAST 22 steps only
ARGS="--steps 20 --warmup 2 --skip-steps 2500"
run stats_16M_iterating_window2500 --kernel-trace --stats
run fetch_16M_window2500 --pmc FETCH_SIZE --kernel-trace
run write_16M_window2500 --pmc WRITE_SIZE --kernel-trace
run sq_16M_iterating_window2500 $SQ --kernel-trace
cd $GRAFT_REPO_ROOT
if [ "$MODE" != bench ]; then
S=tools/summarize_profile.py
for n in stats_1M stats_16M stats_1M_iterating_fixed32 stats_1M_iterating_window3750 stats_16M_iterating_window2500; do
  f=$(find $out/$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $S $f > $out/$n.txt
  f=$(find $out/$n -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/gaps.py $f > $out/gaps_${n#stats_}.txt
done
# (the timed window of the --skip-steps 2500 run: its last 22 steps x 9 launches — per-kernel averages of the window, not of the whole run)
f=$(find $out/stats_16M_iterating_window2500 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/gaps.py $f $(( $(wc -l < $f) - 1 - 198 )) > $out/stats_16M_window2500_last22steps.txt
for n in fetch_1M write_1M fetch_16M write_16M; do f=$(find $out/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $S $f > $out/$n.txt; done
for n in fetch_16M_window2500 write_16M_window2500; do f=$(find $out/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $S $f --last 198 > $out/$n.txt; done
for n in sq_1M sq_16M sq_16M_iterating_window2500; do f=$(find $out/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 tools/pmc_table.py $f > $out/$n.txt; done
HEAD=$(cat $GRAFT_REPO_ROOT/GIT_HEAD.txt 2>/dev/null)
SRC="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of python3 bench.py"
python3 tools/make_traffic_json.py $out/fetch_1M.txt $out/write_1M.txt 999698 $out/traffic_1M.json "$SRC --particles 1000000), profiles/${P}_{fetch,write}_1M.txt; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)" "$HEAD" > /dev/null
python3 tools/make_traffic_json.py $out/fetch_16M.txt $out/write_16M.txt 15995168 $out/traffic_16M.json "$SRC, the default 16 M workload), profiles/${P}_{fetch,write}_16M.txt; FETCH_SIZE doubled per MI355X_MICROARCH.md" "$HEAD" > /dev/null
python3 tools/make_traffic_json.py $out/fetch_16M_window2500.txt $out/write_16M_window2500.txt 15995168 $out/traffic_16M_window2500.json "$SRC --skip-steps 2500, the last 198 dispatches = the window's 22 steps), profiles/${P}_{fetch,write}_16M_window2500.txt; FETCH_SIZE doubled per MI355X_MICROARCH.md" "$HEAD" 2500 > /dev/null
VS="rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ... --kernel-trace (python3 bench.py"
python3 tools/make_valu_json.py $out/sq_1M.txt 999698 $out/valu_1M.json "$VS --particles 1000000), profiles/${P}_sq_1M.txt" "$HEAD" > /dev/null
python3 tools/make_valu_json.py $out/sq_16M.txt 15995168 $out/valu_16M.json "$VS, the default 16 M workload), profiles/${P}_sq_16M.txt" "$HEAD" > /dev/null
fi
if [ "$MODE" = prof ]; then find $out -name "*.csv" -delete; find $out -type d -empty -delete; exit 0; fi
# bench lines (with roofline + cpu_baseline) of the same build
b() { name=$1; shift; timeout 900 python3 bench.py "$@" > $out/bench_$name.json 2> $out/bench_$name.err; echo "bench $name rc=$?"; }
b default --steps 20 --warmup 5
b 1M --steps 100 --particles 1000000 --no-cpu-baseline --no-also
b 1M_fixed32 --steps 100 --particles 1000000 --fixed-iterations 3 2 --no-cpu-baseline --no-also
b 1M_window3750 --steps 200 --particles 1000000 --skip-steps 3750 --no-cpu-baseline --no-also
b 16M_window2500 --steps 20 --warmup 2 --skip-steps 2500 --no-cpu-baseline --no-also
SPHX_FUSE_DIV=0 b 1M_no_fused_divergence --steps 100 --particles 1000000 --no-cpu-baseline --no-also
b 1M_wcsph --steps 100 --particles 1000000 --solver wcsph --no-cpu-baseline --no-also
b 64M --steps 5 --warmup 1 --particles 64000000 --no-cpu-baseline --no-roofline --no-also
b 128M --steps 5 --warmup 1 --particles 128000000 --no-cpu-baseline --no-roofline --no-also
b 1M_forcetiles --steps 100 --particles 1000000 --force-tiles --no-cpu-baseline --no-roofline --no-also
b 16M_forcetiles --steps 20 --warmup 2 --force-tiles --no-cpu-baseline --no-roofline --no-also
find $out -name "*.csv" -delete
find $out -type d -empty -delete
ls $out
