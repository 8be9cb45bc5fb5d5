cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -20
for i in 1 2; do
python bench.py --no-cpu-baseline --no-roofline --steps 400 --force-tiles | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('tiles', d['ms_per_step'], d['value']/1e9)"
python bench.py --no-cpu-baseline --no-roofline --steps 400 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('single', d['ms_per_step'], d['value']/1e9)"
done
