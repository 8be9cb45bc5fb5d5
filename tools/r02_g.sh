cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_harness.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/g_pytest.txt
cat gpurun_out/g_pytest.txt
S=15.713   # sqrt(1e6/4050)
for i in 1 2; do
./yasph2d_amd/sphx_harness --solver dfsph --scale $S --steps 400 --warmup 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('harness', d['particles'], d['particle_steps_per_s']/1e9, d['particles']/d['particle_steps_per_s']*1e3)"
python bench.py --no-cpu-baseline --no-roofline --steps 400 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench', d['config']['particles_total'], d['ms_per_step'], d['value']/1e9)"
done
