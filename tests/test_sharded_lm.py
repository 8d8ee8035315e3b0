"""N>1 path on CPU: world_size-2 gloo run of the sharded LM loop must take exactly the control path --
and produce exactly the results -- of the unsharded run.  The kernels are a TEST stand-in
built on the oracle (tests/oracle_kernels.py); what is under test is theseus_amd's host logic: batch
sharding, the batch-global predicates through DistBatchReducer, the solution all_gather."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.helpers import golden_problem, load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _take(g, idx):
    """Golden fixture restricted / re-ordered to the batch items ``idx``."""
    g = dict(g)
    for k in ("poses0", "meas", "prior_target"):
        g[k] = g[k][idx]
    for k in ("w_between", "w_prior"):
        if g[k].shape[0] > 1:
            g[k] = g[k][idx]
    return g


def _load(name, perm):
    g = load_golden(name)
    return g if perm is None else _take(g, list(perm))


def _run_lm(g, opt_overrides, reducer=None, spy=None):
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from tests.test_gpu_lm import build_objective
    _, _, kw = golden_problem(g)
    kw = dict(kw)
    kw.pop("gauss_newton")
    obj, _ = build_objective(th, g, device="cpu")
    okw = dict(max_iterations=kw.pop("max_iterations"), step_size=kw.pop("step_size"))
    okw.update(opt_overrides)
    cls = th.Dogleg if kw.pop("dogleg", False) else th.LevenbergMarquardt
    opt = cls(obj, linearization_kwargs=dict(kernels=OracleKernels()), **okw)
    if reducer is not None:
        opt.reducer = reducer
    if spy is not None:
        inner = opt.reducer.decide

        def decide(any_f, all_f):
            r = inner(any_f, all_f)
            spy.append(([bool(f.any()) for f in any_f], [bool(f.all()) for f in all_f], r))
            return r
        opt.reducer.decide = decide
        inner_all = opt.reducer.device_all

        def device_all(flag):   # the sync-free loop's form of the same predicate (0-dim device bool)
            r = inner_all(flag)
            spy.append(([], [bool(flag)], ([], [bool(r)])))
            return r
        opt.reducer.device_all = device_all
    sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(track_err_history=True, **kw))
    packed = opt.linear_solver.linearization.packed
    final = torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1)
    return final, info, packed


def _worker(rank, world, port, name, perm, overrides, outdir):
    import torch.distributed as dist
    from theseus_amd.sharding import DistBatchReducer, gather_solution, shard_bounds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = _load(name, perm)
        lo, hi = shard_bounds(g["poses0"].shape[0], rank, world)
        spy = []
        final, info, packed = _run_lm(_take(g, slice(lo, hi)), overrides, reducer=DistBatchReducer(), spy=spy)
        gathered = gather_solution(packed.tensors.poses)  # (P, B_total, 3, 4) on every rank
        local_all_rejected = any(len(loc_all) > 0 and loc_all[0] and not res[1][0] for _, loc_all, res in spy)
        torch.save(dict(final=final, err=info.err_history, iters=info.iters_done,
                        status=[s.name for s in info.status], gathered=gathered,
                        local_all_rejected=local_all_rejected), os.path.join(outdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


# In pg_f64_lm_adaptive_rejects the first LM step is rejected for problems 1, 2, 3, 5 and accepted for 0, 4:
# with the batch ordered (1, 2, 3 | 0, 4, 5) rank 0's shard is ALL rejected while the global batch is not.
PERM = (1, 2, 3, 0, 4, 5)
CASES = [
    ("pg_f64_lm_adaptive_rejects", PERM, dict(abs_err_tolerance=0.0, rel_err_tolerance=0.0)),
    ("pg_f64_lm_adaptive_rejects", None, dict(abs_err_tolerance=1e-9, rel_err_tolerance=1e-7)),
    ("pg_f64_lm_adaptive", None, dict(abs_err_tolerance=1e-10, rel_err_tolerance=1e-3)),
    ("pg_f64_lm", None, dict(abs_err_tolerance=0.0, rel_err_tolerance=0.0)),
    ("pg2_f64_lm_adaptive", None, dict(abs_err_tolerance=0.0, rel_err_tolerance=0.0)),   # SE2, uneven shards (B=5)
    # Dogleg: "every Gauss-Newton step lies inside its trust region" (dogleg.py:55-58) is a batch-global predicate too
    ("pg_f64_dogleg", None, dict(abs_err_tolerance=0.0, rel_err_tolerance=0.0)),
    ("pg_f64_dogleg_rejects", None, dict(abs_err_tolerance=0.0, rel_err_tolerance=0.0)),
]


def _spawn(world, name, perm, overrides):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), name, perm, overrides, d), nprocs=world, join=True)
        return [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=False) for r in range(world)]


@pytest.mark.parametrize("name,perm,overrides", CASES)
def test_two_rank_sharded_lm_equals_unsharded(name, perm, overrides):
    world = 2
    ref_final, ref_info, _ = _run_lm(_load(name, perm), overrides)
    outs = _spawn(world, name, perm, overrides)
    if perm is not None:
        # this case exercises a batch-global predicate: a shard whose problems are ALL rejected while the
        # global batch is not -- without the all-reduce that rank would take the all-rejected retry path
        assert any(o["local_all_rejected"] for o in outs)
    got = torch.cat([o["final"] for o in outs], 0)
    np.testing.assert_allclose(got.numpy(), ref_final.numpy(), rtol=0, atol=1e-12)
    assert all(o["iters"] == ref_info.iters_done for o in outs)           # same control path on every rank
    k = ref_info.iters_done + 1
    np.testing.assert_allclose(torch.cat([o["err"] for o in outs], 0)[:, :k].numpy(),
                               ref_info.err_history[:, :k].numpy(), rtol=1e-10)  # batched MKL kernels round differently per batch size
    assert sum((o["status"] for o in outs), []) == [s.name for s in ref_info.status]
    # the one data-path collective: every rank holds every shard's solution
    for o in outs:
        np.testing.assert_allclose(o["gathered"].transpose(0, 1).numpy(), ref_final.numpy(), rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["pg_f64_lm_adaptive_ellips", "pg2_f64_lm_adaptive", "pg2_f64_lm", "pg3_f64_lm", "pg3_f64_lm_adaptive"])
def test_reference_trajectory_with_standin_kernels(name):
    """The stand-in + the host LM loop reproduce the REAL reference's recorded trajectory: the host loop is
    the reference's control flow (this is what the GPU tests check with the HIP kernels plugged in)."""
    g = load_golden(name)
    final, info, _ = _run_lm(g, dict(abs_err_tolerance=0.0, rel_err_tolerance=0.0))
    np.testing.assert_allclose(final.numpy(), g["final"], rtol=0, atol=5e-8)


def _fail_worker(rank, world, port, outdir):
    """Gauss-Newton (sync-free loop): the problems of rank 1 are singular from the start, rank 0's are fine."""
    import warnings
    import torch.distributed as dist
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    from tests.test_gpu_lm import build_objective
    from theseus_amd.sharding import DistBatchReducer, shard_bounds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = dict(load_golden("pg_f64_gn"))
        lo, hi = shard_bounds(g["poses0"].shape[0], rank, world)
        g = _take(g, slice(lo, hi))
        if rank == 1:
            g["w_between"], g["w_prior"] = g["w_between"] * 0.0, g["w_prior"] * 0.0
        obj, _ = build_objective(th, g, device="cpu")
        opt = th.GaussNewton(obj, linearization_kwargs=dict(kernels=OracleKernels()), max_iterations=3,
                             abs_err_tolerance=0.0, rel_err_tolerance=0.0)
        opt.reducer = DistBatchReducer()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sol, info = th.TheseusLayer(opt).forward(None, optimizer_kwargs={})
        final = torch.stack([sol[f"pose_{k}"] for k in range(int(g["P"]))], 1)
        torch.save(dict(final=final, start=torch.from_numpy(g["poses0"]), status=[s.name for s in info.status],
                        iters=info.iters_done), os.path.join(outdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_a_failure_on_one_shard_freezes_every_shard():
    """nonlinear_least_squares.py:138-152 on a sharded batch, sync-free loop: the solve of ONE shard fails at the first
    iteration -> every rank reports FAIL and no rank has moved its variables (what the unsharded reference run does)."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_fail_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        outs = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=False) for r in range(2)]
    for o in outs:
        assert all(s == "FAIL" for s in o["status"]) and o["iters"] == 0
        assert torch.equal(o["final"], o["start"])


def test_rccl_smoke_script_logic_over_gloo():
    """tests/rccl_smoke.py is what the >= 2-GPU test runs over RCCL; here the same script over gloo on the CPU (uneven shards
    through gather_solution, the reducer's predicates)."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(THX_SMOKE_BACKEND="gloo", PYTHONPATH=root + os.pathsep + env.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "rccl_smoke.py")],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL-SMOKE-OK" in out.stdout, out.stderr[-3000:]
