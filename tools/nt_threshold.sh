cd $GRAFT_REPO_ROOT
for P in 2000000 4000000 8000000; do
 for r in 1 2; do
  for v in 0 1; do
   SPHX_NT_COLD_STORES=$v python3 bench.py --no-cpu-baseline --no-roofline --no-also --steps 100 --warmup 10 --particles $P 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('P=$P nt=$v', round(d['ms_per_step']*1000,1), 'us/step', round(d['value']/1e9,3), 'G/s')"
  done
 done
done
