cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_random_scenes.py tests/test_golden.py tests/test_gpu_harness.py tests/test_gpu_wcsph.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -20
for a in "" "--skip-steps 3750"; do
python bench.py --no-cpu-baseline --no-roofline --steps 400 $a | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$a', d['ms_per_step'], d['value']/1e9)"
done
python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 2 --particles 16000000 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('16M', d['ms_per_step'], d['value']/1e9)"
