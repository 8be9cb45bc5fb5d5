#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_tiles_full.py tests/test_gpu_multi.py -x -q -k "invariant" > $out/pytest.log 2>&1; echo "rc=$?"; tail -15 $out/pytest.log
