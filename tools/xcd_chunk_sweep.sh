cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-also $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step']*1000,1), 'us/step', round(d['value']/1e9,3), 'G/s')"; }
for r in 1 2; do
 for v in 5 7 9 11; do
  export SPHX_XCD_CHUNK=$v
  run "shift=$v 16M late" "--steps 20 --warmup 2 --skip-steps 2500"
  run "shift=$v 16M" "--steps 60 --warmup 10"
 done
 for v in 4 5 6 7; do
  export SPHX_XCD_CHUNK=$v
  run "shift=$v 1M late" "--steps 200 --warmup 5 --particles 1000000 --skip-steps 3750"
  run "shift=$v 1M" "--steps 300 --warmup 20 --particles 1000000"
 done
done
