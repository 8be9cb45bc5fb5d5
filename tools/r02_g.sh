cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -20
for i in 1 2; do
for e in 1 0; do
SPHX_FUSE_DIV=$e python bench.py --no-cpu-baseline --no-roofline --steps 200 --skip-steps 3750 --force-tiles | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('fuse_div=$e tiles late', d['ms_per_step'], d['value']/1e9)"
done; done
