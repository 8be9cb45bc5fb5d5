#!/bin/bash
# rocprofv3 passes on the GPU box:  tools/r02_prof.sh OUTNAME [extra bench args...]   (1M by default)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline $*"
run() { name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d $out/$name -- $B $ARGS > $out/$name.log 2>&1; echo "$name rc=$?"; }
ARGS="--steps 100 --warmup 5"
run stats --kernel-trace --stats
ARGS="--steps 20 --warmup 2"
run fetch --pmc FETCH_SIZE --kernel-trace
run write --pmc WRITE_SIZE --kernel-trace
run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace
run sq2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace
cd $GRAFT_REPO_ROOT
S=tools/summarize_profile.py
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 $S $f > $out/stats.txt
for n in fetch write; do f=$(find $out/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $S $f > $out/$n.txt; done
for n in sq sq2 tcc; do f=$(find $out/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 tools/pmc_table.py $f > $out/$n.txt; done
f=$(find $out/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/gaps.py $f > $out/gaps.txt
find $out -name "*.csv" -size +1M -delete
cat $out/stats.txt; cat $out/sq.txt; cat $out/sq2.txt; cat $out/fetch.txt; cat $out/write.txt; cat $out/tcc.txt
