#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_bench_launcher.py tests/test_gpu_multi.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
b() { name=$1; shift; timeout 600 "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']
    print(sys.argv[2], 'n_gpus=%d'%d['n_gpus'], 'G=%.3f'%(d['value']/1e9), 'ms=%.4f'%d['ms_per_step'], c['parallelism'][:160])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
b single_1M python bench.py --steps 200 --no-cpu-baseline --no-roofline
b tile1_1M python bench.py --steps 200 --no-cpu-baseline --no-roofline --force-tiles
b tile1_rccl_1M env SPHX_RCCL_ALWAYS=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 200 --no-cpu-baseline --no-roofline --force-tiles
b gloo2_500k python bench.py --gpus 2 --backend gloo --particles 500000 --steps 100 --no-cpu-baseline --no-roofline
b gloo4_250k python bench.py --gpus 4 --backend gloo --particles 250000 --steps 100 --no-cpu-baseline --no-roofline
b gloo8_125k python bench.py --gpus 8 --backend gloo --particles 125000 --steps 100 --no-cpu-baseline --no-roofline
tail -3 $out/gloo8_125k.err
