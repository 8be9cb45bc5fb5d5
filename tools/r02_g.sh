cd $GRAFT_REPO_ROOT
for i in 1 2; do
for e in 0 1; do
SPHX_NO_FUSED_COUNT=$e python bench.py --no-cpu-baseline --no-roofline --steps 400 --force-tiles | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('nofused=$e', d['ms_per_step'], d['value']/1e9)"
done; done
