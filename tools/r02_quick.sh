#!/bin/bash
# quick GPU check: parity tests (fast subset or all) + the 1M / 16M bench lines with the per-kernel table
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ "$1" = all ]; then T="tests"; else T="tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_random_scenes.py tests/test_gpu_wcsph.py tests/test_golden.py"; fi
timeout 1200 python -m pytest $T -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2], 'G=%.3f'%(d['value']/1e9), 'ms=%.4f'%d['ms_per_step'], 'Id=%.2f Iv=%.2f k=%.2f'%(c['mean_density_iterations'],c['mean_divergence_iterations'],c['mean_neighbors'] or 0))
    print('   ', {k[:22]:round(v*1000,1) for k,v in d['roofline']['per_kernel_ms_per_step_event_inflated'].items()})
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
b 1M --steps 200
b 16M --steps 20 --warmup 2 --particles 16000000
b 1M_late --steps 200 --skip-steps 3750
