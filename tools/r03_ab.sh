#!/bin/bash
# A/B on one box: parity subset with the product library, then bench lines (1M, 16M, late window) for base and each variant.
#   tools/r03_ab.sh OUTNAME [all|fast|none] VARIANT...
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mode=$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ "$mode" = all ]; then T="tests"; elif [ "$mode" = fast ]; then T="tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_random_scenes.py tests/test_golden.py"; else T=""; fi
if [ -n "$T" ]; then timeout 1500 python -m pytest $T -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log; fi
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2], 'G=%.3f'%(d['value']/1e9), 'ms=%.4f'%d['ms_per_step'], 'Id=%.2f Iv=%.2f k=%.2f'%(c['mean_density_iterations'],c['mean_divergence_iterations'],c['mean_neighbors'] or 0))
    print('   ', {k[:26]:round(v*1000,1) for k,v in d['roofline']['per_kernel_ms_per_step_event_inflated'].items()})
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
for v in base "$@"; do
  if [ $v = base ]; then unset SPHX_LIB; else export SPHX_LIB=$PWD/yasph2d_amd/variants/libsphx_$v.so; fi
  b 1M_$v --steps 200
  b 16M_$v --steps 20 --warmup 2 --particles 16000000
  b 1M_late_$v --steps 200 --skip-steps 3750
done
