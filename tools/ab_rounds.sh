cd $GRAFT_REPO_ROOT   # needs the round-5 tree beside this one: mkdir r05tree && git archive 83b6c60 | tar -x -C r05tree && make -C r05tree/yasph2d_amd/csrc
run() { (cd $1 && python3 bench.py --no-cpu-baseline --no-roofline --no-also $3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', round(d['ms_per_step']*1000,1), 'us/step', round(d['value']/1e9,3), 'G/s')"); }
for r in 1 2 3; do
  run r05tree "r05 16M" "--steps 100 --warmup 10"
  run . "r06 16M" "--steps 100 --warmup 10"
done
for r in 1 2; do
  run r05tree "r05 1M" "--steps 200 --warmup 20 --particles 1000000"
  run . "r06 1M" "--steps 200 --warmup 20 --particles 1000000"
  run r05tree "r05 16M late" "--steps 20 --warmup 2 --skip-steps 2500"
  run . "r06 16M late" "--steps 20 --warmup 2 --skip-steps 2500"
done
