#!/bin/bash
# GPU call 1 of round 2: baseline of the round-1 build on this round's box + where the iterating regime starts
out=$GRAFT_REPO_ROOT/gpurun_out/r02_probe1
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_bench_launcher.py tests/test_gpu_parity.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"
timeout 300 python bench.py --steps 100 --no-cpu-baseline > $out/bench_1M.json 2> $out/bench_1M.err; echo "bench rc=$?"
timeout 300 python tools/iter_trace.py --particles 1000000 --steps 4500 --every 100 > $out/iter_trace_1M.txt 2> $out/iter_trace_1M.err; echo "trace rc=$?"
timeout 200 python tools/iter_trace.py --particles 250000 --steps 4000 --every 100 > $out/iter_trace_250k.txt 2> $out/iter_trace_250k.err; echo "trace rc=$?"
# does RCCL accept two ranks on ONE GPU?  (decides how the multi-rank RCCL path can be exercised on a 1-GPU box)
cat > /tmp/rccl_probe.py <<'PY'
import os, torch, torch.distributed as dist
r = int(os.environ["RANK"]); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.full((4,), float(r + 1), device="cuda")
dist.all_reduce(t); print("allreduce", r, t.tolist(), flush=True)
a = torch.full((8,), float(r), device="cuda"); b = torch.empty(8, device="cuda")
ops = [dist.P2POp(dist.isend, a, 1 - r), dist.P2POp(dist.irecv, b, 1 - r)]
[w.wait() for w in dist.batch_isend_irecv(ops)]
torch.cuda.synchronize(); print("p2p", r, b.tolist(), flush=True)
dist.destroy_process_group()
PY
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 /tmp/rccl_probe.py > $out/rccl_same_gpu.log 2>&1; echo "rccl probe rc=$?"
tail -5 $out/rccl_same_gpu.log
ls /opt/rocm/lib | grep -i rccl | head; ls /opt/rocm/include/rccl 2>/dev/null | head
python - <<'PY'
import torch, os
print(torch.__file__)
import glob
print(glob.glob(os.path.dirname(torch.__file__) + "/lib/*rccl*"))
PY
