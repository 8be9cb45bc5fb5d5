cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_random_scenes.py tests/test_golden.py tests/test_gpu_harness.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -20
for i in 1 2; do
for e in 1 0; do
SPHX_FUSE_DIV=$e python bench.py --no-cpu-baseline --no-roofline --steps 200 --skip-steps 3750 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('fuse_div=$e late', d['ms_per_step'], d['value']/1e9)"
done; done
for e in 1 0; do
SPHX_FUSE_DIV=$e python bench.py --no-cpu-baseline --no-roofline --steps 100 --fixed-iterations 3 2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('fuse_div=$e fixed32', d['ms_per_step'], d['value']/1e9)"
done
