cd $GRAFT_REPO_ROOT
for v in fake1 fake2 fake3; do
  unset SPHX_LIB
  [ $v != base ] && export SPHX_LIB=$PWD/yasph2d_amd/variants/libsphx_$v.so
  timeout 120 python bench.py --no-cpu-baseline --no-also --steps 200 --skip-steps 3750 2>/dev/null | python -c "
import sys,json
try:
  d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); pk=d['roofline']['per_kernel_ms_per_step_event_inflated']
  print('$v', round(d['ms_per_step'],4), {k[:30]:round(x*1000,1) for k,x in pk.items() if 'compute_density_error' in k or 'prediction' in k})
except Exception as e: print('$v failed', e)"
done
