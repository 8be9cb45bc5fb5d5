for h in 16 8 6; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29658 bench.py --gpus 2 --steps 60 --warmup 5 --particles 500000 --backend gloo --no-roofline --halo $h 2>/dev/null > gpurun_out/halo_$h.json
  python - gpurun_out/halo_$h.json $h <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read())
print("halo", sys.argv[2], round(d["value"] / 1e9, 3), round(d["ms_per_step"], 4), d["config"]["parallelism"][-70:])
PY
done
