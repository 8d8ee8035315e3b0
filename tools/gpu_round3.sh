#!/bin/bash
# Round 3's measurement set in ONE gpurun call (outputs under gpurun_out/<tag>/; copy what is to be judged into profiles/r3/):
# the GPU test suite, the default bench.py line (headline + all legs) the way the driver runs it, and the bundle-adjustment leg at
# 5 / 10 / 20 iterations per optimize() (per-solve kernel time vs once-per-optimize host work).
set -u
TAG=${1:-r3z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -s > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tail -3 $OUT/bench_default.time; wc -c $OUT/bench_default.json
python - > $OUT/ba_iterations_sweep.txt 2>&1 <<'PY'
import sys, json, torch
sys.path.insert(0, ".")
import bench
from types import SimpleNamespace
ctx = SimpleNamespace(world=1, rank=0, device=torch.device("cuda", 0), on_gpu=True, kernels=None, dist=None)
torch.cuda.set_device(0)
for steps in (5, 10, 20, 40):
    r = bench.ba_run(SimpleNamespace(cams=512, points=8192, batch=256, dtype="f32", steps=steps, warmup=2, parity=False, cpu_baseline=False), ctx)
    it = r["roofline"]["iteration"]
    print(f"max_iterations {steps:3d}: accepted {r['iters_done']:3d}, linear solves {r['linear_solves']:3d}, wall per optimize {r['ms_per_step'] * r['iters_done']:8.2f} ms, "
          f"per solve {it['ms_per_solve']:6.2f} ms (kernels {it['kernel_ms_per_solve']:6.2f}), value {r['value']:8.0f} problem-iterations/s")
PY
grep -v amdgpu $OUT/ba_iterations_sweep.txt
