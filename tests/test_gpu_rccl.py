"""-m gpu, needs >= 2 GPUs (skipped on the 1-GPU boxes): the sharded path's collectives over RCCL itself -- two ranks, one GPU
each, uneven shards (tests/rccl_smoke.py) -- and `bench.py --gpus 2` end to end with the real kernels (weak headline + the
strong-scaling leg in ONE line)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_two = pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@needs_two
def test_rccl_collectives_of_the_sharded_path():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "rccl_smoke.py")],
                         cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL-SMOKE-OK" in out.stdout, out.stderr[-3000:]


@needs_two
def test_bench_two_gpus_weak_headline_and_strong_leg():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "256", "--legs", "strong", "--strong-total", "1024"], cwd=ROOT, env=_env(),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["collective_backend"] == "nccl" and r["all_gather_ms"] is not None
    leg = r["configs"]["strong_f64_1024"]
    assert leg["scaling"] == "strong" and leg["config"]["global_batch"] == 1024 and leg["dtype"] == "f64"
    assert leg["mean_error"][1] < leg["mean_error"][0]
