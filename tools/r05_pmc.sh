#!/bin/bash
# GPU box: counter passes at 16 M (or ARGS) for the question "what binds the walks".  tools/r05_pmc.sh OUTNAME [bench args...]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0 --steps 10 --warmup 2 $@"
pass() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- $B > $out/$name.log 2>&1; echo "$name rc=$?"
  f=$(find $out/$name -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $f > $out/$name.txt; find $out/$name -name "*.csv" -delete; find $out -type d -empty -delete; }
pass lds SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS
pass vmem SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
# (the derived TCC_*_sum / TA_*_sum counters did not finish within 400 s per pass on this pool: not collected)
head -12 $out/lds.txt $out/vmem.txt
