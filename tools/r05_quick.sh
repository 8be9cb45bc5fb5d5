#!/bin/bash
# GPU box: parity subset + bench at 1 M (+ SQ counters of the neighbour build).  tools/r05_quick.sh OUTNAME [full]
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ "$2" = full ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
else
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_scenes.py tests/test_gpu_edges.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
fi
for P in 1000000 16000000; do
timeout 300 python bench.py --steps $([ $P = 1000000 ] && echo 100 || echo 20) --particles $P --no-cpu-baseline --no-also > $out/bench_$P.json 2> $out/bench_$P.err; echo "bench rc=$?"
python3 - $out/bench_$P.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e9,3), round(d['ms_per_step'],4), {k[:30]:round(v*1000,1) for k,v in d['roofline']['per_kernel_ms_per_step_event_inflated'].items()})
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/sq -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-also --prewarm-ms 0 --steps 20 --warmup 2 --particles 1000000 > $out/sq.log 2>&1; echo "sq rc=$?"
f=$(find $out/sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $f > $out/sq_1M.txt
find $out/sq -name "*.csv" -delete; find $out -type d -empty -delete
cat $out/sq_1M.txt | head -20
