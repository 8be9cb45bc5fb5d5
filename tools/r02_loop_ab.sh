#!/bin/bash
# device-run vs host-run solver loops: parity tests + timing of the iterating regime (GPU box)
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2], 'G=%.3f'%(d['value']/1e9), 'ms=%.4f'%d['ms_per_step'], 'Id=%.2f Iv=%.2f W=%s'%(c['mean_density_iterations'],c['mean_divergence_iterations'],c['warmstart_rate']))
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
for mode in dev host; do
  if [ $mode = host ]; then export SPHX_HOST_LOOP=1; else unset SPHX_HOST_LOOP; fi
  b ${mode}_1M_adaptive --steps 200
  b ${mode}_1M_fixed11 --steps 200 --fixed-iterations 1 1
  b ${mode}_1M_fixed21 --steps 200 --fixed-iterations 2 1
  b ${mode}_1M_fixed32 --steps 200 --fixed-iterations 3 2
  b ${mode}_1M_fixed55 --steps 200 --fixed-iterations 5 5
  b ${mode}_1M_tol --steps 200 --skip-steps 300 --tolerance-scale 0.01
  b ${mode}_1M_win3800 --steps 200 --skip-steps 3750
  b ${mode}_250k_win1200 --steps 600 --skip-steps 1200 --particles 250000
  b ${mode}_16M_fixed32 --steps 20 --warmup 2 --fixed-iterations 3 2 --particles 16000000
  b ${mode}_16M_fixed11 --steps 20 --warmup 2 --fixed-iterations 1 1 --particles 16000000
  b ${mode}_4k_adaptive --steps 1000 --skip-steps 200 --particles 4050
done
