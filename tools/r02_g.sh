cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_random_scenes.py tests/test_gpu_tiles.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head -20
for a in ; do
python bench.py --no-cpu-baseline --steps 200 $a | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$a', d['ms_per_step'], d['value']/1e9, {k[:12]:round(v*1000,1) for k,v in d['roofline']['per_kernel_ms_per_step_event_inflated'].items()})"
done
