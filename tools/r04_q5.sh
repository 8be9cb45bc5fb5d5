#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_scenes.py tests/test_gpu_edges.py tests/test_gpu_tiles.py tests/test_gpu_api_fuzz.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
show() { python3 - $1 <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e9,3), round(d['ms_per_step'],4), {k[:34]:round(v*1000,1) for k,v in d['roofline']['per_kernel_ms_per_step_event_inflated'].items()})
PY
}
timeout 300 python bench.py --steps 100 --particles 1000000 --no-cpu-baseline --no-also > $out/b1.json 2>$out/b1.err; show $out/b1.json
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-also > $out/b16.json 2>$out/b16.err; show $out/b16.json
timeout 600 python bench.py --steps 200 --particles 1000000 --skip-steps 3750 --no-cpu-baseline --no-also > $out/blate.json 2> $out/blate.err; show $out/blate.json
