#!/bin/bash
# Pre-flight of bench.py's multi-rank path with the REAL kernels on a 1-GPU box: two ranks share cuda:0 (THX_BENCH_ONE_DEVICE=1),
# collectives over gloo.  What it cannot show: RCCL itself (one device per rank is what RCCL wants).
mkdir -p gpurun_out/preflight
export THX_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 \
  --backend gloo --steps 3 --warmup 1 --batch 1024 --strong-total 4096 --cpu-sample 0 --parity-sample 0 > gpurun_out/preflight/line.json 2> gpurun_out/preflight/err.txt
echo "exit $?"; tail -3 gpurun_out/preflight/err.txt | cut -c1-300
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/preflight/line.json") if l.startswith("{")][-1])
print("n_gpus", r["n_gpus"], "ranks", r.get("ranks"), "backend", r.get("collective_backend"), "value", round(r["value"]), "scaling", r["scaling"],
      "all_gather_ms", r.get("all_gather_ms"), "legs", {k: round(v["value"]) for k, v in r.get("configs", {}).items()})
PY
