"""bench.py's launcher / sharding / reporting logic without a GPU: `python bench.py --gpus 2` must START two ranks by itself
(torch.distributed.run, 127.0.0.1 rendezvous), shard the batch, run the one all_gather and print ONE JSON line that says so.
The kernels are the TEST stand-in (tests/oracle_kernels.py) over gloo -- the line is marked "data": "TEST-STANDIN"; on the GPU
box the same command runs libtheseus_hip.so over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEAM = ["--backend", "gloo", "--test-kernels", "tests.oracle_kernels:OracleKernels", "--poses", "6", "--edges", "8",
        "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--parity-sample", "0", "--dtype", "f64"]


def _bench(*argv, seam=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv, *(SEAM if seam is None else seam)], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 prints ONE line
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_weak_scaling():
    r = _bench("--gpus", "2", "--batch", "3")
    assert r["n_gpus"] == 2 and r["ranks"] == 2 and r["collective_backend"] == "gloo"
    assert r["scaling"] == "weak" and r["config"]["global_batch"] == 6 and r["config"]["batch_per_gpu"] == 3
    assert r["all_gather_ms"] is not None and r["all_gather_ms"] >= 0.0
    assert r["rank_ms_per_step"]["min"] <= r["rank_ms_per_step"]["max"] <= r["ms_per_step"] * 1.001
    assert r["iters_done"] == 2 and r["data"] == "TEST-STANDIN" and "roofline" not in r
    assert r["value"] == pytest.approx(6 * 2 / (r["ms_per_step"] * 2 * 1e-3), rel=1e-6)
    assert r["mean_error"][1] < r["mean_error"][0]


def test_gpus_2_strong_scaling_in_sub_batches():
    r = _bench("--gpus", "2", "--total-batch", "8", "--batch", "2")
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert r["config"]["global_batch"] == 8 and r["config"]["batch_per_gpu"] == 4 and "2 sub-batches of 2" in r["config"]["parallelism"]


def test_single_rank_default():
    r = _bench("--batch", "2")
    assert r["n_gpus"] == 1 and r["ranks"] == 1 and r["all_gather_ms"] is None and r["collective_backend"] is None


def test_gpus_2_with_the_default_rank0_legs_does_not_hang():
    """The driver runs `bench.py --gpus N --steps K --warmup W` with nothing else: the rank-0-only legs (CPU baseline, parity
    sub-sample) must not run the sharded loop's collectives on rank 0 alone while the other ranks sit in the final barrier --
    at N > 1 they are skipped (they belong to the N = 1 line)."""
    seam = [a for a in SEAM if a not in ("--cpu-sample", "--parity-sample", "0")] + ["--cpu-iters", "1"]
    r = _bench("--gpus", "2", "--batch", "3", seam=seam)
    assert r["n_gpus"] == 2 and "cpu_baseline" not in r and "parity" not in r
    r1 = _bench("--batch", "3", seam=seam)
    assert r1["n_gpus"] == 1 and r1["cpu_baseline"]["kind"] == "port" and "parity" in r1


def test_gpus_2_emits_the_strong_scaling_leg_in_the_same_line():
    """At N > 1 the driver's ONE command must also yield BASELINE.json configs[2] (a fixed job sharded over the ranks, sub-batches,
    all_gather): the leg rides under "configs" next to the weak-scaling headline, every rank takes part in it."""
    r = _bench("--gpus", "2", "--batch", "2", "--legs", "strong", "--strong-total", "8")
    assert r["scaling"] == "weak" and r["config"]["global_batch"] == 4
    leg = r["configs"]["strong_f64_8"]
    assert leg["scaling"] == "strong" and leg["n_gpus"] == 2 and leg["config"]["global_batch"] == 8
    assert leg["config"]["batch_per_gpu"] == 4 and "2 sub-batches of 2" in leg["config"]["parallelism"]
    assert leg["all_gather_ms"] is not None and leg["iters_done"] == 2
    assert leg["rank_ms_per_step"]["min"] <= leg["rank_ms_per_step"]["max"]


def test_single_rank_legs_ride_in_the_same_line():
    r = _bench("--batch", "2", "--legs", "fp64")
    assert r["n_gpus"] == 1 and r["configs"]["fp64_b4096"]["dtype"] == "f64" and r["configs"]["fp64_b4096"]["iters_done"] == 2
    assert "configs" not in _bench("--batch", "2")          # a modified headline run carries no legs by default


def test_config0_leg_rides_in_the_line():
    """BASELINE.json configs[0] (examples/simple_example.py shape) as a leg: generic path, implicit backward, parity against the closed
    form of the linear least-squares problem -- no oracle involved."""
    r = _bench("--batch", "2", "--legs", "simple")
    leg = r["configs"]["simple_example_b16"]
    assert leg["parity"]["max_abs_v_err"] < 1e-12 and leg["parity"]["grad_x_rel_err"] < 1e-9 and leg["iters_done"] >= 1


def test_gpus_8_is_the_drivers_command_shape():
    """The driver's 8-GPU line is `bench.py --gpus 8 --steps K --warmup W` and nothing else: eight ranks, the weak-scaling headline
    (every rank its own batch, one all_gather of eight shards) AND the strong-scaling leg in the same line, in the shape of
    configs[2] -- 32768 / 8 = 4096 per rank in sub-batches -- scaled down: 32 problems, 4 per rank, 2 sub-batches of 2, every rank
    in every collective.  Stand-in kernels over gloo: what is checked is that the path is correct by construction -- it cannot
    hang, it shards what it says, its all_gather returns world x batch problems."""
    r = _bench("--gpus", "8", "--batch", "2", "--legs", "strong", "--strong-total", "32")
    assert r["n_gpus"] == 8 and r["ranks"] == 8 and r["collective_backend"] == "gloo" and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 16 and r["config"]["batch_per_gpu"] == 2 and r["all_gather_ms"] is not None
    assert r["iters_done"] == 2 and r["value"] == pytest.approx(16 * 2 / (r["ms_per_step"] * 2 * 1e-3), rel=1e-6)
    leg = r["configs"]["strong_f64_32"]
    assert leg["scaling"] == "strong" and leg["n_gpus"] == 8 and leg["config"]["global_batch"] == 32
    assert leg["config"]["batch_per_gpu"] == 4 and "2 sub-batches of 2" in leg["config"]["parallelism"]
    assert leg["all_gather_ms"] is not None and leg["iters_done"] == 2


def test_gpus_8_uneven_shards():
    """A job that does not divide by the rank count: 37 problems over 8 ranks = shares of 5, 5, 5, 5, 5, 4, 4, 4 (shard_bounds),
    each within one sub-batch -- the all_gather carries unequal shards, the batch-global predicates reduce over all 37."""
    r = _bench("--gpus", "8", "--total-batch", "37", "--batch", "8")
    assert r["n_gpus"] == 8 and r["scaling"] == "strong" and r["config"]["global_batch"] == 37
    assert r["value"] == pytest.approx(37 * 2 / (r["ms_per_step"] * 2 * 1e-3), rel=1e-6)
    assert r["iters_done"] == 2 and r["all_gather_ms"] is not None
