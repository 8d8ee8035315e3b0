#!/usr/bin/env python
"""bench.py -- LM problem-iterations/s on the synthetic SE3 pose graph (BASELINE.json configs[1]) + every other BASELINE config
as a leg of the same JSON line.

A "step" is ONE Levenberg-Marquardt iteration over the whole batch (linearize -> damp + Cholesky
factor + solve -> retract -> error), i.e. one pass of the hot path over one batch of synthetic
input.  `--steps K` LM iterations are run by a single TheseusLayer.forward (max_iterations=K, both
tolerances 0 so nothing exits early -- same trick as the reference's examples/pose_graph/pose_graph_cube.py:95-96);
the timed region is bracketed by barrier + torch.cuda.synchronize and the max over ranks is reported.
value = n_gpus * B * K / time  (problem-iterations per second, whole job).

The headline (`metric` / `value` / `roofline` / `cpu_baseline` / `parity` at the top level) is BASELINE.json configs[1]:
256 poses / 1024 edges, batch 4096 per GPU, fp32, LM + dense Cholesky.  Under `"configs"` the same line carries
  N = 1 : "fp64_b4096"       configs[2]'s dtype and per-GPU share (32768 / 8 problems) on one GPU,
          "ba_512_8192_32768_b256"   configs[3] (bundle adjustment, Schur complement + tile-sparse Cholesky),
          "implicit_b1024"   configs[4] (forward LM + implicit backward through TheseusLayer),
          "simple_example_b16"   configs[0] (the plumbing case: AutoDiffCostFunction on a Vector, Gauss-Newton + implicit
                             backward on the generic path; reports correctness against the closed form, ~1 s),
  N > 1 : "strong_f64_32768" configs[2] itself: 32768 fp64 problems sharded over the N GPUs (strong scaling), each rank's
          share solved in sub-batches of 4096, one all_gather of the solved poses,
each with its own value / ms_per_step / roofline / parity / cpu_baseline (`--legs none` skips them).

Multi GPU (one rank per GPU; `python bench.py --gpus N` re-executes itself under torch.distributed.run when it was not
launched by it): the batch dimension shards -- every rank owns B independent problems (weak scaling) -- and one RCCL
all_gather re-collects the solved poses on every rank inside the timed region.  No other collective.
"""
import argparse
import gc
import importlib
import json
import os
import socket
import subprocess
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {"f32": 157.3, "f64": 78.6}  # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


class KernelTimer:
    """HIP-event timing of every C-ABI call, on the stream the kernels are launched on
    (torch's current stream: torch.cuda.Event records there)."""

    NAMES = ["pg_assemble", "pg_assemble_blocks", "chol_factor", "chol_factor_sparse", "chol_factor_hblocks", "chol_factor_levels",
             "chol_solve_backward", "chol_solve", "chol_solve_sparse", "chol_solve_levels",
             "se3_retract", "pg_error", "lm_accept", "ba_assemble", "ba_schur", "ba_schur_blocks", "ba_backsub", "ba_error", "ba_retract",
             "vec_gather"]

    def __init__(self, K):
        self.K, self.events, self.enabled = K, {}, False
        for n in self.NAMES:
            if hasattr(K, n):
                self.events[n] = []
                setattr(K, n, self._wrap(n, getattr(K, n)))

    def _wrap(self, name, fn):
        def wrapped(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.events[name].append((e0, e1))
            return r
        return wrapped

    def summary(self):
        out = {}
        for n, ev in self.events.items():
            if ev:
                ms = [a.elapsed_time(b) for a, b in ev]
                out[n] = {"calls": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms)}
        return out


def oracle_problem(tensors, edges, P, dtype, sample, first=0):
    from oracle import pose_graph as opg
    from theseus_amd.utils import synthetic as syn
    cpu = lambda t: t[first:first + sample].detach().cpu().to(dtype)  # noqa: E731
    poses0 = torch.stack([cpu(tensors[f"VERTEX_SE3__{k}"]) for k in range(P)], 1)
    meas = torch.stack([cpu(tensors[f"EDGE_SE3__{i}_{j}"]) for (i, j) in edges], 1)
    E = len(edges)
    w = torch.tensor([1 / syn.TRANSLATION_NOISE] * 3 + [1 / syn.ROTATION_NOISE] * 3, dtype=dtype)
    prob = opg.PGProblem(num_poses=P, edges=torch.tensor(edges), meas=meas, w_between=w.view(1, 1, 6).expand(1, E, 6),
                         prior_idx=torch.tensor([0]), prior_target=cpu(tensors["VERTEX_SE3__0__PRIOR"]).unsqueeze(1),
                         w_prior=torch.full((1, 1, 6), syn.PRIOR_WEIGHT, dtype=dtype))
    return prob, poses0


def cpu_baseline(tensors, edges, P, dtype, sample, iters, damping, chunk):
    """Oracle (CPU restatement of the reference's dense algorithm, torch-CPU/MKL) on a bounded sample of the same workload,
    in chunks (dense A is 37.8 MB (fp32) / 75.6 MB (fp64) per problem: SURVEY §8d runs the reference's CPU path in chunks of
    64 / 32).  Returns (problem-iters/s, cores, final poses of the sample, seconds, cost history)."""
    from oracle import pose_graph as opg
    finals, hists, dt = [], [], 0.0
    with torch.no_grad():
        for first in range(0, sample, chunk):
            prob, poses0 = oracle_problem(tensors, edges, P, dtype, min(chunk, sample - first), first)
            t0 = time.perf_counter()
            final, info = opg.lm_optimize(prob, poses0, max_iterations=iters, damping=damping, abs_err_tolerance=0.0,
                                          rel_err_tolerance=0.0)
            dt += time.perf_counter() - t0
            finals.append(final)
            hists.append(torch.stack(info.err_history, 1))
    return sample * iters / dt, torch.get_num_threads(), torch.cat(finals), dt, torch.cat(hists)


REFERENCE_RECORDED = {   # the unmodified reference on 8 host threads of the build container (where /root/reference exists)
    "value": 12.58, "unit": "problem-iterations/s", "cores": 8,
    "what": "theseus LevenbergMarquardt + DenseLinearization + CholeskyDenseSolver (vectorize=True), 256 problems x 3 LM "
            "iterations in chunks of 64 at the headline size, fp32; the oracle port on the same data and threads: 10.10",
    "source": "profiles/r3/k_reference_proper_cpu_timing.txt (tools/reference_cpu_timing.py)"}


def reference_root():
    """Where an importable copy of the reference lives (THX_REFERENCE_ROOT, else /root/reference), or None: the GPU box has
    none -- the line then carries the oracle port as its CPU baseline plus the recorded figure of the real reference."""
    for r in (os.environ.get("THX_REFERENCE_ROOT"), "/root/reference"):
        if r and os.path.isdir(os.path.join(r, "theseus")):
            return r
    return None


def reference_cpu_baseline(tensors, edges, P, dtype, sample, iters, damping, chunk):
    """SURVEY.md 8(d) / BASELINE.md 3: the UNMODIFIED reference -- theseus.LevenbergMarquardt + DenseLinearization +
    CholeskyDenseSolver, vectorize=True, torch-CPU on all host threads -- on the first ``sample`` problems of the batch, in chunks.
    Returns (problem-iters/s, threads, final poses of the sample, seconds) or None when no copy of the reference imports."""
    root = reference_root()
    if root is None:
        return None
    try:
        for p_ in (os.path.join(ROOT, "oracle", "stubs"), root, os.path.join(root, "torchlie"), os.path.join(root, "torchkin")):
            if p_ not in sys.path:
                sys.path.insert(0, p_)
        import warnings
        warnings.filterwarnings("ignore")
        import theseus as rth
    except Exception as e:   # a half-staged copy must not take the line down
        print(f"[bench] the reference at {root} does not import ({type(e).__name__}: {e}); CPU baseline = the oracle port", file=sys.stderr)
        return None
    from theseus_amd.utils import synthetic as syn
    cpu = lambda t, a, b: t[a:b].detach().cpu().to(dtype)  # noqa: E731
    finals, dt = [], 0.0
    for first in range(0, sample, chunk):
        last = min(first + chunk, sample)
        obj = rth.Objective(dtype=dtype)
        poses = [rth.SE3(tensor=cpu(tensors[f"VERTEX_SE3__{k}"], first, last), name=f"VERTEX_SE3__{k}") for k in range(P)]
        info = torch.tensor([[1 / syn.TRANSLATION_NOISE] * 3 + [1 / syn.ROTATION_NOISE] * 3], dtype=dtype)
        weight = rth.DiagonalCostWeight(rth.Variable(info, name="EDGE_WEIGHT"))
        for (i, j) in edges:    # examples/pose_graph/pose_graph_synthetic.py:130-152
            meas = rth.SE3(tensor=cpu(tensors[f"EDGE_SE3__{i}_{j}"], first, last), name=f"EDGE_SE3__{i}_{j}")
            obj.add(rth.Between(poses[i], poses[j], meas, weight, name=f"between_{i}_{j}"))
        target = rth.SE3(tensor=cpu(tensors["VERTEX_SE3__0__PRIOR"], first, last), name="VERTEX_SE3__0__PRIOR")
        pw = rth.ScaleCostWeight(rth.Variable(torch.tensor([[syn.PRIOR_WEIGHT]], dtype=dtype), name="PRIOR_WEIGHT"))
        obj.add(rth.Difference(poses[0], target, pw, name="pose_prior"))
        opt = rth.LevenbergMarquardt(obj, linear_solver_cls=rth.CholeskyDenseSolver, vectorize=True, max_iterations=iters,
                                     step_size=1.0, abs_err_tolerance=0.0, rel_err_tolerance=0.0)
        with torch.no_grad():
            t0 = time.perf_counter()
            opt.optimize(damping=damping)
            dt += time.perf_counter() - t0
        finals.append(torch.stack([p_.tensor for p_ in poses], 1))
    return sample * iters / dt, torch.get_num_threads(), torch.cat(finals), dt


def relative_poses(X):
    """Gauge-free view of a solution (B, P, 3, 4): X_k^-1 X_{k+1} along the odometry chain."""
    from oracle import lie
    B = X.shape[0]
    return lie.se3_compose(lie.se3_inverse(X[:, :-1].reshape(-1, 3, 4)), X[:, 1:].reshape(-1, 3, 4)).view(B, -1, 3, 4)


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start N ranks, one per GPU, and hand their output through."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


class exact_thresholds:
    """Context: the fp64 oracle evaluates with the RUN dtype's Taylor thresholds -- the exact evaluation of what an fp32 run
    (the reference's or ours) approximates."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        from oracle import lie
        self.saved = dict(lie.EPS[torch.float64])
        if self.dtype == torch.float32:
            import numpy as np
            lie.EPS[torch.float64] = {k: float(np.float32(v)) for k, v in lie.EPS[torch.float32].items()}

    def __exit__(self, *a):
        from oracle import lie
        lie.EPS[torch.float64] = self.saved


def exact_reference(tensors, edges, P, dtype, sample, iters, damping):
    """The same problem evaluated exactly: fp64 oracle on the (fp32- or fp64-valued) inputs with the
    run dtype's Taylor thresholds -- what both the reference's fp path and ours approximate."""
    from oracle import pose_graph as opg
    prob, poses0 = oracle_problem(tensors, edges, P, torch.float64, sample)
    with exact_thresholds(dtype), torch.no_grad():
        final, info = opg.lm_optimize(prob, poses0, max_iterations=iters, damping=damping, abs_err_tolerance=0.0,
                                      rel_err_tolerance=0.0)
    return final, torch.stack(info.err_history, 1)


def riemannian(X, G):
    """Tangent projection of a gradient w.r.t. the raw 3x4 entries of X: [R skew(R^T G_R) | G_t] -- the component normal to
    SO(3) depends on how a closed form extends off the manifold (Taylor vs exact branch), so only the projected gradient
    compares across dtypes (tests/test_gpu_implicit.py)."""
    R = X[..., :3]
    M = R.transpose(-1, -2) @ G[..., :3]
    return torch.cat([R @ (0.5 * (M - M.transpose(-1, -2))), G[..., 3:]], -1)


def chain_relative(X):
    """(B, P, 3, 4) -> X_k^-1 X_{k+1} (B, P-1, 3, 4) in plain differentiable torch ops (any device): the gauge-free view of a
    solution.  The parity gradients are taken of sum(chain_relative(final)): at this size the undamped Gauss-Newton system of
    the implicit step has cond ~ 6e14 (the prior of weight 1e-3 is all that pins the gauge), so a loss that sees the gauge
    has gradients no two evaluations agree on (tests/implicit_common.py:check_full_size_implicit)."""
    R0, t0, R1, t1 = X[:, :-1, :, :3], X[:, :-1, :, 3:], X[:, 1:, :, :3], X[:, 1:, :, 3:]
    Rt = R0.transpose(-1, -2)
    return torch.cat([Rt @ R1, Rt @ (t1 - t0)], -1)


class limited_threads:
    """torch-CPU autograd through thousands of small ops is op-overhead bound: on a 128-core host the intra-op thread pool makes
    it several times SLOWER than 8 threads (measured: 102 s vs ~20 s for two problems)."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.saved = torch.get_num_threads()
        torch.set_num_threads(min(self.n, self.saved))
        return torch.get_num_threads()

    def __exit__(self, *a):
        torch.set_num_threads(self.saved)


def on_manifold(t):
    """(…, 3, 4) SE3 tensors with fp32-rounded rotations -> the nearest rotation in fp64 (polar projection)."""
    t = t.double().clone()
    U, _, Vh = torch.linalg.svd(t[..., :3])
    t[..., :3] = U @ Vh
    return t


def oracle_implicit(tensors, edges, P, dtype, sample, iters, damping, exact):
    """Forward LM (iters - 1 iterations, no grad) + the grad-enabled Gauss-Newton step + backward of the gauge-free loss
    sum(chain_relative(final poses)) through the oracle (oracle.pose_graph.implicit_final_step: autograd through the restated formulas, pinned to the
    reference's implicit gradients by tests/test_oracle_golden.py).  ``exact``: fp64 with the run dtype's thresholds.
    Returns (final poses, d loss / d measurements, seconds)."""
    import contextlib
    import dataclasses
    from oracle import pose_graph as opg
    prob, poses0 = oracle_problem(tensors, edges, P, torch.float64 if exact else dtype, sample)
    ctx = exact_thresholds(dtype) if exact else contextlib.nullcontext()
    t0 = time.perf_counter()
    with ctx:
        with torch.no_grad():
            x, _ = opg.lm_optimize(prob, poses0, max_iterations=iters - 1, damping=damping, abs_err_tolerance=0.0,
                                   rel_err_tolerance=0.0)
        meas = prob.meas.clone().requires_grad_(True)
        final, _ = opg.implicit_final_step(dataclasses.replace(prob, meas=meas), x, fallback_damping=damping)
        chain_relative(final).sum().backward()
    return final.detach(), meas.grad, time.perf_counter() - t0


def implicit_fixture(P, E, dtype, iters, damping):
    """tests/golden/bench_implicit_exact.npz (tools/gen_implicit_parity_fixture.py): four problems of the implicit leg's workload
    with their EXACT forward + implicit step + gradients (fp64 oracle), so that the leg's parity needs no ~150 s of torch-CPU
    autograd per bench run.  None when the leg runs at another size / dtype / iteration count: the exact evaluation then runs live."""
    path = os.path.join(ROOT, "tests", "golden", "bench_implicit_exact.npz")
    if dtype != torch.float32 or not os.path.exists(path):
        return None
    import numpy as np
    z = np.load(path)
    if int(z["P"]) != P or int(z["E"]) != E or int(z["cpu_iters"]) != iters or abs(float(z["damping"]) - damping) > 1e-15:
        return None
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    edges = [tuple(e) for e in z["edges"].tolist()]
    tensors = {f"VERTEX_SE3__{k}": t("poses0")[:, k].contiguous() for k in range(P)}
    tensors.update({f"EDGE_SE3__{i}_{j}": t("meas")[:, n].contiguous() for n, (i, j) in enumerate(edges)})
    tensors["VERTEX_SE3__0__PRIOR"] = t("prior")
    return {"tensors": tensors, "problems": int(z["problems"]), "edges": edges, "seed": int(z["seed"]),
            "ex_final": t("ex_final"), "ex_grad": t("ex_grad"), "ex64_final": t("ex64_final"), "ex64_grad": t("ex64_grad")}


# Algorithmic HBM bytes per problem of the HBM-bound kernels: n = 6 P columns, E edges, element size es.
def algorithmic_bytes(P, E, es):
    n = 6 * P
    rec = 12 * es
    return {
        # reads poses + measurements, writes the E + P (+ prior) lower 6x6 blocks and g  (SURVEY §8d counts the dense lower
        # triangle here; the kernel writes only the blocks of the fixed pattern -- since round 3 as a block LIST)
        "pg_assemble": (P + E + 1) * rec + (E + P) * 36 * es + n * es,
        "pg_error": (P + E + 1) * rec,
        "se3_retract": 2 * P * rec + n * es,
        "chol_solve_backward": n * (n + 1) // 2 * es + 2 * n * es,   # tril(L) once + y + x
    }


def compact(x, digits=9):
    """Floats to ``digits`` significant digits, recursively: the line carries five configurations and must stay readable (and
    inside whatever tail of stdout a harness keeps)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if x == x and abs(x) != float("inf") else x
    if isinstance(x, dict):
        return {k: compact(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [compact(v, digits) for v in x]
    return x


def free_device_memory():
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def pg_run(cfg, ctx):
    """One pose-graph measurement (the headline, or a leg): build the objective, W warm-up + K timed LM iterations, the one
    all_gather at N > 1; on rank 0: the result dict with roofline / cpu_baseline / parity.  Returns None on other ranks."""
    import theseus_amd as th
    from theseus_amd.utils import synthetic as syn
    world, rank, device, on_gpu, kernels, dist = ctx.world, ctx.rank, ctx.device, ctx.on_gpu, ctx.kernels, ctx.dist
    standin = kernels is not None
    dtype = torch.float32 if cfg.dtype == "f32" else torch.float64
    P, E, B, K_iters, W = cfg.poses, cfg.edges, cfg.batch, cfg.steps, cfg.warmup
    n = 6 * P
    edges = syn.pose_graph_topology(P, E, topology_seed=0)
    objective = syn.build_pose_graph_objective(edges, P, dtype=dtype, device=device)
    solver_cls = th.HipSparseCholeskySolver if cfg.solver == "sparse" else th.HipCholeskySolver
    opt = th.LevenbergMarquardt(objective, linear_solver_cls=solver_cls, max_iterations=K_iters,
                                abs_err_tolerance=0.0, rel_err_tolerance=0.0, step_size=1.0,
                                linearization_kwargs=dict(kernels=kernels) if standin else None)
    if world > 1:
        from theseus_amd.sharding import DistBatchReducer
        opt.reducer = DistBatchReducer()  # batch-global predicates over all shards (one tiny all-reduce / iteration)
    layer = th.TheseusLayer(opt)
    timer = KernelTimer(opt.linear_solver.K)
    strong = cfg.total_batch > 0
    if strong:
        from theseus_amd.sharding import plan_sub_batches
        if cfg.implicit:
            raise SystemExit("--total-batch is the forward configuration (configs[2]); not combined with --implicit")
        try:
            B, n_sub = plan_sub_batches(cfg.total_batch, rank, world, B)
        except ValueError as e:
            raise SystemExit(str(e))
    else:
        n_sub = 1
    # every sub-batch's inputs are resident in HBM before the timed region (synthetic, one seed per rank and sub-batch)
    sub_inputs = []
    for c in range(n_sub):
        tensors_c = syn.make_pose_graph_tensors(edges, P, B, dtype=dtype, device=device, seed=1234 + rank + 1000 * c,
                                                kernels=kernels)
        sub_inputs.append(syn.input_dict(tensors_c))
        if c == 0:
            tensors = tensors_c
    inputs = sub_inputs[0]
    okw = dict(damping=cfg.damping, adaptive_damping=cfg.adaptive)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
            sync()

    bwd_ms, gather_ms = None, None
    if cfg.implicit:
        for k, v in inputs.items():
            if k.startswith("EDGE_SE3__"):
                v.requires_grad_(True)
        okw = dict(okw, backward_mode="implicit")
    with torch.set_grad_enabled(cfg.implicit):
        if W > 0:   # (the same optimizer kwargs as the timed call: every kernel of the timed path -- the history writes at
                    #  device-side indices included -- has been loaded and run once before the clock starts)
            opt.set_params(max_iterations=max(W, 2) if cfg.implicit else W)
            sol_w, _ = layer.forward(inputs, optimizer_kwargs=dict(track_err_history=True, **okw))
            if cfg.implicit:   # ... and the backward's kernels
                torch.stack(list(sol_w.values())).sum().backward()
                for v in inputs.values():
                    v.grad = None
            del sol_w
        opt.set_params(max_iterations=K_iters)
        barrier()
        timer.enabled = on_gpu
        t0 = time.perf_counter()
        sol, info = layer.forward(inputs, optimizer_kwargs=dict(track_err_history=True, **okw))
        for more in sub_inputs[1:]:  # strong scaling: the rank's remaining sub-batches through the same workspaces
            layer.forward(more, optimizer_kwargs=okw)
        if cfg.implicit:  # backward: retract VJP + ONE linear solve with the cached factor + cost VJP
            loss = torch.stack(list(sol.values())).sum()   # (one reduction over all poses; 256 separate .sum() calls were
                                                            #  1.6 ms of 6 us kernels in the timed backward)
            eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            eb0.record()
            loss.backward()
            eb1.record()
            torch.cuda.synchronize()
            bwd_ms = eb0.elapsed_time(eb1)
        if world > 1:  # the one data-path collective: re-collect the solved poses on every rank (RCCL over xGMI)
            from theseus_amd.sharding import gather_solution
            sync()
            tg0 = time.perf_counter()
            gathered = gather_solution(opt.linear_solver.linearization.packed.tensors.poses)
            sync()
            gather_ms = (time.perf_counter() - tg0) * 1e3
            # (weak scaling: world x B problems; strong scaling: the last sub-batch of every rank's share -- shares of a job that
            #  does not divide by the rank count differ by one, shard_bounds)
            assert gathered.shape[1] == (world * B if not strong else sum(
                plan_sub_batches(cfg.total_batch, r_, world, cfg.batch)[0] for r_ in range(world))), gathered.shape
            del gathered
        local_dt = time.perf_counter() - t0   # this rank's own work, before it waits for the others
        barrier()
        dt = time.perf_counter() - t0
        timer.enabled = False
    rank_ms = [local_dt * 1e3]
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        every = [None] * world
        dist.all_gather_object(every, (local_dt * 1e3, gather_ms))
        rank_ms = [e[0] for e in every]
        gather_ms = max(e[1] for e in every)

    iters_done = info.iters_done
    result = None
    if rank == 0:
        phases = timer.summary()
        es = 4 if cfg.dtype == "f32" else 8
        err_hist = info.err_history
        result = {
            "metric": "LM iterations/sec (batch x vars) on SE3 pose-graph",
            "value": (cfg.total_batch if strong else world * n_sub * B) * iters_done / dt,
            "unit": "problem-iterations/s",
            "n_gpus": world, "steps": K_iters, "warmup": W, "ms_per_step": dt / max(iters_done, 1) * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": cfg.dtype, "data": "synthetic" if on_gpu else "TEST-STANDIN",
            "config": {"workload": f"SE3 pose-graph {P} poses / {E} Between edges + 1 prior, batch {n_sub * B} per GPU, "
                                   f"LM damping {cfg.damping}{' adaptive' if cfg.adaptive else ''} + "
                                   f"{'tile-sparse Cholesky (RCM ordering)' if cfg.solver == 'sparse' else 'dense Cholesky'}",
                       "poses": P, "edges": E, "batch_per_gpu": n_sub * B, "global_batch": cfg.total_batch if strong else world * n_sub * B,
                       "n": n,
                       "parallelism": f"batch-shard x{world}" + (f", {n_sub} sub-batches of {B} per GPU" if strong else "")},
            "ranks": world if world == 1 else dist.get_world_size(),
            "collective_backend": None if world == 1 else dist.get_backend(),
            "rank_ms_per_step": {"min": min(rank_ms) / max(iters_done, 1), "max": max(rank_ms) / max(iters_done, 1)},
            "all_gather_ms": gather_ms,
            "pose_updates_per_s": (cfg.total_batch if strong else world * n_sub * B) * iters_done * P / dt,
            "iters_done": iters_done,
            "mean_error": [float(err_hist[:, 0].mean()), float(err_hist[:, iters_done].mean())],
        }
        if on_gpu:
            sparse = cfg.solver == "sparse"
            fac = (phases.get("chol_factor_hblocks") or phases.get("chol_factor_sparse" if sparse else "chol_factor")
                   or {"avg_ms": float("nan")})
            if "pg_assemble_blocks" in phases:    # (block-compact Hessian: the same role in the tables below)
                phases["pg_assemble"] = phases["pg_assemble_blocks"]
            # n^3/3 flops per problem x B problems per thx_chol_factor_forward call (the 2n^2 of the
            # fused forward substitution are not counted)
            dense_flops = B * (n ** 3) / 3.0
            peak = PEAK[cfg.dtype]
            # the tile-sparse solver EXECUTES fewer flops than the dense count: its matrix-core utilisation is executed flops /
            # time / peak (``achieved`` / ``frac``); the dense-equivalent figure is reported beside it, not as the fraction
            pat = opt.linear_solver.pattern if sparse else None
            flops_per_launch = B * pat.flops if sparse else dense_flops
            achieved = flops_per_launch / (fac["avg_ms"] * 1e-3) / 1e12
            traffic, traffic_src = None, None
            try:  # measured offline with rocprofv3 --pmc (bench.py cannot profile itself): profiles/traffic.json
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(
                    f"{cfg.dtype}_n{n}_b{B}" + ("_sparse" if sparse else ""))
                if tj:
                    traffic, traffic_src = tj["bytes_per_factor_call"], tj["source"]
            except (OSError, ValueError, KeyError):
                pass
            # the HBM-bound kernels of the iteration: algorithmic bytes / HIP-event time
            hbm = {}
            for name, per_problem in algorithmic_bytes(P, E, es).items():
                if name in phases:
                    gbs = per_problem * B / (phases[name]["avg_ms"] * 1e-3) / 1e9
                    hbm[name] = {"avg_ms": round(phases[name]["avg_ms"], 4), "algorithmic_GBps": round(gbs, 1),
                                 "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
            # SURVEY §8(d) per problem-iteration totals: n^3/3 + 2n^2 flops, 5 n^2/2-ish bytes (23.7 MB fp32 at n = 1536)
            t_mfma = B * (n ** 3 / 3.0 + 2.0 * n * n) / (peak * 1e12) * 1e3
            t_hbm = B * (4 * n * (n + 1) / 2 * es + (P + E + 1) * 12 * es + 2 * n * es) / (HBM_PEAK_GBS * 1e9) * 1e3
            step_ms = dt / max(iters_done, 1) * 1e3 / n_sub
            result["roofline"] = {
                "bound": "mfma",
                "kernel": ("thx_chol_factor_hblocks" + (" along the tile pattern" if sparse else "") if "chol_factor_hblocks" in phases
                           else ("thx_chol_factor_sparse" if sparse else "thx_chol_factor_forward")) +
                          " (chol_syrk + chol_potrf [or chol_diag] + chol_offdiag launches per block column; fp32 dense frames: chol_offdiag2 per column pair)",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": traffic, "traffic_unit": "bytes per factor call (PMC, rocprofv3)",
                "traffic_source": traffic_src, "flops_per_launch": flops_per_launch, "avg_launch_ms": fac["avg_ms"],
                "hbm_bound_kernels": hbm,
                "executed": None if not sparse else {
                    "of_dense": pat.flops / pat.dense_flops,
                    "tiles_of_L": [pat.l_tiles, pat.ntiles * (pat.ntiles + 1) // 2],
                    "dense_equivalent_TFLOPs": dense_flops / (fac["avg_ms"] * 1e-3) / 1e12},
                "iteration": {"floor_ms": {"mfma": round(t_mfma, 3), "hbm": round(t_hbm, 3)}, "ms_per_step": step_ms,
                              "frac": max(t_mfma, t_hbm) / step_ms}}
            result["phases_ms_per_call"] = {k: round(v["avg_ms"], 4) for k, v in phases.items()}
        S, CI = min(cfg.cpu_sample, B), cfg.cpu_iters
        SP = min(cfg.parity_sample, B)
        if world > 1:
            # the CPU baseline and the parity sub-sample are rank-0-only legs: at N > 1 the other ranks are already waiting in the
            # final barrier, and the parity run would issue the sharded loop's all-reduces alone.  They belong to the N = 1 line.
            S = SP = 0
        cpu_final = cpu_hist = None
        if cfg.implicit:
            result["config"]["workload"] += " + implicit backward through TheseusLayer"
            result["implicit_backward_ms"] = bwd_ms
            if on_gpu:
                result["roofline"]["note"] = ("a step = one forward LM iteration; the grad-enabled Gauss-Newton step and the "
                                              "backward (retract VJP + one solve with the cached factor + cost VJP) are inside the "
                                              "timed region and divided over the forward iterations")
            port_final = port_grad = None
            # the leg's CPU work runs on the four FIXTURE problems of this workload (same generator, another seed) whose exact
            # evaluation is cached (implicit_fixture): the port is timed on them -- the leg's cpu_baseline -- and is the fp32 band
            # the HIP gradients are read against; at any other size the exact evaluation runs live on the batch's first problems
            fx = implicit_fixture(P, E, dtype, CI, cfg.damping) if (S > 0 or SP > 0) else None
            ptensors, psrc = tensors, "first {n} problems of the batch"
            if fx is not None and list(fx["edges"]) == [tuple(e) for e in edges]:
                ptensors, psrc = fx["tensors"], f"{{n}} fixture problems of this workload (seed {fx['seed']}, tests/golden/bench_implicit_exact.npz)"
                S, SP = min(S, fx["problems"]), min(SP, fx["problems"])
            else:
                fx = None
            if S > 0:
                with limited_threads(8) as nthr:
                    port_final, port_grad, cpu_s = oracle_implicit(ptensors, edges, P, dtype, S, CI, cfg.damping, exact=False)
                v = S * CI / cpu_s
                result["cpu_baseline"] = {"value": v, "unit": "problem-iterations/s", "cores": nthr,
                                          "kind": "port",
                                          "sample": psrc.format(n=S) + f", {CI - 1} LM iterations + the implicit "
                                                    f"Gauss-Newton step + backward ({cpu_s:.1f} s), oracle.pose_graph (torch-CPU "
                                                    f"autograd through the restated formulas)",
                                          "reference": "the reference's implicit mode at this size is recorded as a fixture "
                                                       "(tests/golden/pg_full_f64_implicit.npz, oracle/gen_golden.py: its gradients "
                                                       "pin this port and the HIP path), not timed"}
                result["speedup_vs_cpu"] = result["value"] / v
            if SP > 0:
                if fx is not None:
                    ex_final, ex_grad = fx["ex_final"][:SP], fx["ex_grad"][:SP]
                else:
                    with limited_threads(8):
                        ex_final, ex_grad, _ = oracle_implicit(ptensors, edges, P, dtype, SP, CI, cfg.damping, exact=True)
                sub = {k: ptensors[k][:SP].detach().to(device).clone().requires_grad_(k.startswith("EDGE_SE3__")) for k in inputs}
                opt.set_params(max_iterations=CI)
                with torch.enable_grad():
                    sol_s, _ = layer.forward(sub, optimizer_kwargs=okw)
                    final_s = torch.stack([sol_s[f"VERTEX_SE3__{k}"] for k in range(P)], 1)
                    chain_relative(final_s).sum().backward()
                got = final_s.detach().cpu().double()
                ggrad = torch.stack([sub[f"EDGE_SE3__{i}_{j}"].grad for (i, j) in edges], 1).cpu().double()
                X = torch.stack([sub[f"EDGE_SE3__{i}_{j}"].detach() for (i, j) in edges], 1).cpu().double()
                gr, er = riemannian(X, ggrad), riemannian(X, ex_grad)
                result["parity"] = {
                    "reference": "fp64 oracle (exact evaluation of the same inputs): forward + implicit step + autograd backward "
                                 "of the gauge-free loss sum(X_k^-1 X_{k+1}); gradients compared in the tangent projection"
                                 + ("; cached: tests/golden/bench_implicit_exact.npz (tools/gen_implicit_parity_fixture.py)" if fx is not None else ""),
                    "problems": SP, "iters": CI,
                    "tolerance": "fp32: the implicit step is the UNDAMPED Gauss-Newton solve of a system with cond ~ 6e14 (the 1e-3 "
                                 "prior is all that pins the gauge): no fp32 evaluation resolves it -- read hip_* against cpu_port_* "
                                 "(the same algorithm in fp32 on the CPU); the code path's parity statement is f64_rerun "
                                 "(gradients 1e-4, gauge-free poses 1e-8 against the exact evaluation)",
                    "hip_max_abs_pose_err": float((got - ex_final).abs().max()),
                    "hip_max_rel_pose_err": float((relative_poses(got) - relative_poses(ex_final)).abs().max()),
                    "hip_grad_meas_rel_err": float((gr - er).abs().max() / er.abs().max()),
                    "grad_scale": float(er.abs().max())}
                if port_final is not None and S >= SP:   # the CPU port in the run dtype: the band an fp32 evaluation lives in
                    pf, pg = port_final[:SP].double(), riemannian(X, port_grad[:SP].double())
                    result["parity"].update({
                        "cpu_port_max_abs_pose_err": float((pf - ex_final).abs().max()),
                        "cpu_port_max_rel_pose_err": float((relative_poses(pf) - relative_poses(ex_final)).abs().max()),
                        "cpu_port_grad_meas_rel_err": float((pg - er).abs().max() / er.abs().max())})
                del sub, sol_s, final_s
                if cfg.dtype == "f32":
                    # The undamped Gauss-Newton system of the implicit step has cond ~ 6e14 at this size (the 1e-3 prior is all
                    # that pins the gauge): NO fp32 evaluation -- the reference's included, see cpu_port_* -- resolves it, the
                    # step's gauge component and the gradients through H^-1 are noise.  The same sub-sample through the HIP path
                    # in fp64 is the parity statement for this configuration's code path (inputs projected onto the manifold in
                    # fp64: fp32-rounded rotations are 6e-8 off it, which this conditioning amplifies into the gauge-free part).
                    obj64 = syn.build_pose_graph_objective(edges, P, dtype=torch.float64, device=device)
                    opt64 = th.LevenbergMarquardt(obj64, linear_solver_cls=th.HipCholeskySolver, max_iterations=CI,
                                                  abs_err_tolerance=0.0, rel_err_tolerance=0.0, step_size=1.0)
                    t64 = {k: on_manifold(t[:SP].detach().cpu()) for k, t in ptensors.items() if t.dim() == 3 and t.shape[-2:] == (3, 4)}
                    sub64 = {k: t64[k].to(device).clone().requires_grad_(k.startswith("EDGE_SE3__")) for k in inputs}
                    with torch.enable_grad():
                        sol64, _ = th.TheseusLayer(opt64).forward(sub64, optimizer_kwargs=okw)
                        final64 = torch.stack([sol64[f"VERTEX_SE3__{k}"] for k in range(P)], 1)
                        chain_relative(final64).sum().backward()
                    if fx is not None:
                        ex64_final, ex64_grad = fx["ex64_final"][:SP], fx["ex64_grad"][:SP]
                    else:
                        with limited_threads(8):
                            ex64_final, ex64_grad, _ = oracle_implicit(t64, edges, P, torch.float64, SP, CI, cfg.damping, exact=True)
                    g64 = torch.stack([sub64[f"EDGE_SE3__{i}_{j}"].grad for (i, j) in edges], 1).cpu()
                    X64 = torch.stack([sub64[f"EDGE_SE3__{i}_{j}"].detach() for (i, j) in edges], 1).cpu()
                    gr64, er64 = riemannian(X64, g64), riemannian(X64, ex64_grad)
                    got64 = final64.detach().cpu()
                    result["parity"]["f64_rerun"] = {
                        "what": "the same sub-sample and code path (forward LM, implicit step, retract VJP + cached-factor solve + "
                                "cost VJP) in fp64 against the fp64 oracle",
                        "hip_max_abs_pose_err": float((got64 - ex64_final).abs().max()),
                        "hip_max_rel_pose_err": float((relative_poses(got64) - relative_poses(ex64_final)).abs().max()),
                        "hip_grad_meas_rel_err": float((gr64 - er64).abs().max() / er64.abs().max())}
                    del sub64, sol64, final64, opt64, obj64
        else:
            if S > 0:
                v, cores, cpu_final, cpu_s, cpu_hist = cpu_baseline(tensors, edges, P, dtype, S, CI, cfg.damping, cfg.cpu_chunk)
                port = {"value": v, "unit": "problem-iterations/s", "cores": cores, "kind": "port",
                        "sample": f"first {S} problems of rank 0's batch in chunks of {min(cfg.cpu_chunk, S)} x "
                                  f"{CI} LM iterations ({cpu_s:.1f} s), oracle.pose_graph.lm_optimize (torch-CPU/"
                                  f"MKL restatement of DenseLinearization + CholeskyDenseSolver)"}
                # the CPU baseline is the reference itself wherever a copy of it imports (the build container,
                # THX_REFERENCE_ROOT); on a box without one, the port -- with the recorded figure of the real reference beside it
                ref = reference_cpu_baseline(tensors, edges, P, dtype, S, CI, cfg.damping, cfg.cpu_chunk) if not standin else None
                if ref is not None:
                    rv, rcores, ref_final, ref_s = ref
                    result["cpu_baseline"] = {
                        "value": rv, "unit": "problem-iterations/s", "cores": rcores, "kind": "reference",
                        "sample": f"first {S} problems of rank 0's batch in chunks of {min(cfg.cpu_chunk, S)} x {CI} LM iterations "
                                  f"({ref_s:.1f} s), the UNMODIFIED reference (theseus LevenbergMarquardt + DenseLinearization + "
                                  f"CholeskyDenseSolver, vectorize=True) imported from {reference_root()}",
                        "port_on_the_same_sample": port,
                        "port_vs_reference_max_abs_pose_diff": float((ref_final.double() - cpu_final.double()).abs().max())}
                    v = rv
                else:
                    result["cpu_baseline"] = dict(port, reference_recorded=REFERENCE_RECORDED,
                                                  reference="absent on this box (no THX_REFERENCE_ROOT, no /root/reference)")
                result["speedup_vs_cpu"] = result["value"] / v
            if SP > 0:
                # parity of the HIP path on a sub-sample, against the exact (fp64) evaluation of the same problem -- for the
                # fp64 path that is the oracle itself (pinned to the reference at this size: tests/golden/pg_full_f64_lm.npz),
                # for fp32 the CPU port's own distance from exact is printed next to it (the fp32 band).
                # *_rel_pose_err is gauge-free (relative poses along the chain): the prior of weight 1e-3 pins the gauge weakly.
                ex_final, ex_hist = exact_reference(tensors, edges, P, dtype, SP, CI, cfg.damping)
                sub = {k: t[:SP].contiguous() for k, t in inputs.items()}
                opt.set_params(max_iterations=CI)
                with torch.no_grad():
                    sol_s, info_s = layer.forward(sub, optimizer_kwargs=dict(track_err_history=True, **okw))
                got = torch.stack([sol_s[f"VERTEX_SE3__{k}"] for k in range(P)], 1).cpu().double()
                rel = lambda h: float(((h.double()[:, CI] - ex_hist[:, CI]).abs() / ex_hist[:, CI].abs()).max())  # noqa
                result["parity"] = {
                    "reference": "fp64 oracle (exact evaluation of the same inputs)", "problems": SP, "iters": CI,
                    "tolerance": ("fp32: inside the reference's own fp32 band (hip_* <= cpu_port_* + fp32 rounding against the "
                                  "exact evaluation; tests/test_gpu_full_size.py) -- north_star's literal 1e-5 vs reference is "
                                  "carried by the fp64 path (2e-7 at this size, configs.fp64_b4096.parity)" if cfg.dtype == "f32"
                                  else "fp64: <= 1e-5 vs the reference's run at this size (measured 2e-7 gauge included, 2e-8 "
                                               "gauge-free: tests/test_gpu_full_size.py against tests/golden/pg_full_f64_lm.npz)"),
                    "hip_max_abs_pose_err": float((got - ex_final).abs().max()),
                    "hip_max_rel_pose_err": float((relative_poses(got) - relative_poses(ex_final)).abs().max()),
                    "hip_rel_err_final_cost": rel(info_s.err_history)}
                if cpu_final is not None and S >= SP:
                    c = cpu_final[:SP].double()
                    result["parity"].update({
                        "cpu_port_max_abs_pose_err": float((c - ex_final).abs().max()),
                        "cpu_port_max_rel_pose_err": float((relative_poses(c) - relative_poses(ex_final)).abs().max()),
                        "cpu_port_rel_err_final_cost": rel(cpu_hist[:SP])})
                del sub, sol_s
    want_sparse_leg = (on_gpu and world == 1 and cfg.solver == "dense" and cfg.sparse_leg and not cfg.implicit and not strong)
    del sol, info, layer, opt, objective, timer, sub_inputs
    if not want_sparse_leg:
        del inputs, tensors
    free_device_memory()
    if want_sparse_leg:
        # ---- the same workload once more with HipSparseCholeskySolver (theseus_amd/sparse.py): same kernels, reverse
        #      Cuthill-McKee variable ordering, structurally zero tiles of L and K-loop blocks skipped.  Reported NEXT TO the
        #      dense headline (value / roofline above are the dense solver's).
        obj2 = syn.build_pose_graph_objective(edges, P, dtype=dtype, device=device)
        opt2 = th.LevenbergMarquardt(obj2, linear_solver_cls=th.HipSparseCholeskySolver, max_iterations=K_iters,
                                     abs_err_tolerance=0.0, rel_err_tolerance=0.0, step_size=1.0,
                                     linear_solver_kwargs=dict(ordering="auto", batch_hint=B))   # (the time model ranks the orders at THIS batch)
        layer2 = th.TheseusLayer(opt2)
        timer2 = KernelTimer(opt2.linear_solver.K)
        with torch.no_grad():
            if W > 0:
                opt2.set_params(max_iterations=W)
                layer2.forward(inputs, optimizer_kwargs=dict(track_err_history=True, **okw))
            opt2.set_params(max_iterations=K_iters)
            torch.cuda.synchronize()
            timer2.enabled = True
            t0 = time.perf_counter()
            sol2, info2 = layer2.forward(inputs, optimizer_kwargs=dict(track_err_history=True, **okw))
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            timer2.enabled = False
        pat = opt2.linear_solver.pattern
        sm2 = timer2.summary()
        fms = (sm2.get("chol_factor_levels") or sm2.get("chol_factor_hblocks") or sm2["chol_factor_sparse"])["avg_ms"]
        result["tile_sparse"] = {
            "solver": "HipSparseCholeskySolver (ordering 'auto' at this batch size: " + str(opt2.linear_solver.ordering_info.get("method")) +
                      ("; level-scheduled" if opt2.linear_solver.levels else "; reverse Cuthill-McKee, column-by-column schedule") + ", tile pattern of L)",
            "value": B * info2.iters_done / dt2, "unit": "problem-iterations/s", "ms_per_step": dt2 / info2.iters_done * 1e3,
            "factor_ms": fms, "tiles_of_L": [pat.l_tiles, pat.ntiles * (pat.ntiles + 1) // 2],
            "executed_flops_of_dense": pat.flops / pat.dense_flops,
            "executed_TFLOPs": B * pat.flops / (fms * 1e-3) / 1e12,
            "executed_frac_of_peak": B * pat.flops / (fms * 1e-3) / 1e12 / PEAK[cfg.dtype],
            "mean_error": [float(info2.err_history[:, 0].mean()), float(info2.err_history[:, info2.iters_done].mean())]}
        del sol2, info2, layer2, opt2, obj2, timer2, inputs, tensors
        free_device_memory()
    return result


def sparse_run(cfg, ctx):
    """The reference's SPARSE sweep regime (evaluations/pose_graph_synthetic.sh:5-12: batch 8 - 256, up to 4096 poses; its
    BaspachoSparseSolver path): a 4096-pose SLAM-like graph (odometry chain + local loop closures, shuffled labels), fp32, LM with
    HipSparseCholeskySolver -- tile-level nested dissection + the LEVEL-SCHEDULED factorisation and solves
    (thx_chol_factor_levels / thx_chol_solve_levels): elimination-tree parallelism inside a problem.  Timed: ``steps`` LM
    iterations (the reference's inner_optim.max_iters = 10) in one TheseusLayer.forward, inputs resident.  Beside it the same
    inputs through the column-by-column schedule under reverse Cuthill-McKee (rounds 3-5), which is also the leg's cross-check:
    two different orderings / factorisations must land on the same poses."""
    import theseus_amd as th
    from theseus_amd.utils import synthetic as syn
    P, B, K_iters, dtype, dev = cfg.poses, cfg.batch, cfg.steps, torch.float32, ctx.device
    edges = syn.chain_graph_topology(P, stride=7, span=5, seed=2)
    K = th.default_kernels()
    gen = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda nn, sc: K.se3_exp(sc * (2 * torch.rand(nn, 6, dtype=dtype, device=dev, generator=gen) - 1))  # noqa: E731
    gt = rnd(B * P, 1.5).view(B, P, 3, 4)
    poses0 = K.se3_compose(gt.reshape(-1, 3, 4), rnd(B * P, 0.05)).view(B, P, 3, 4)
    meas = [K.se3_compose(K.se3_compose(K.se3_inverse(gt[:, i].contiguous()), gt[:, j].contiguous()), rnd(B, 0.01)) for (i, j) in edges]
    start = {f"pose_{k}": poses0[:, k].contiguous() for k in range(P)}
    okw = dict(damping=cfg.damping, track_err_history=True)

    def run(ordering):
        obj = th.Objective(dtype=dtype)
        pv = [th.SE3(tensor=poses0[:, k].clone(), name=f"pose_{k}") for k in range(P)]
        w = th.ScaleCostWeight(torch.tensor(5.0, dtype=dtype, device=dev))
        for k, (i, j) in enumerate(edges):
            obj.add(th.Between(pv[i], pv[j], th.SE3(tensor=meas[k].clone(), name=f"m_{k}"), w, name=f"b_{k}"))
        obj.add(th.Difference(pv[edges[0][0]], th.SE3(tensor=gt[:, edges[0][0]].clone(), name="anchor"), w, name="prior"))
        opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipSparseCholeskySolver, max_iterations=K_iters, abs_err_tolerance=0.0,
                                    rel_err_tolerance=0.0, linear_solver_kwargs=dict(ordering=ordering, batch_hint=B))
        layer = th.TheseusLayer(opt)
        timer = KernelTimer(opt.linear_solver.K)
        with torch.no_grad():
            opt.set_params(max_iterations=max(cfg.warmup, 1))
            layer.forward(start, optimizer_kwargs=okw)
            opt.set_params(max_iterations=K_iters)
            torch.cuda.synchronize()
            # the leg's time: the median of three calls WITHOUT the per-kernel events (they cost host time, and at batch 64 this
            # loop is bound by the host); one more call with them for the phase table
            dts = []
            for _ in range(3):
                t0 = time.perf_counter()
                sol, info = layer.forward(start, optimizer_kwargs=okw)
                torch.cuda.synchronize()
                dts.append(time.perf_counter() - t0)
            dt = sorted(dts)[1]
            timer.enabled = True
            layer.forward(start, optimizer_kwargs=okw)
            torch.cuda.synchronize()
            timer.enabled = False
        ph = timer.summary()
        # (un-wrap the kernels object: the next run builds its own timer)
        for n_ in timer.events:
            setattr(opt.linear_solver.K, n_, getattr(type(opt.linear_solver.K), n_).__get__(opt.linear_solver.K))
        x = torch.stack([sol[f"pose_{k}"] for k in range(P)], 1)
        return dt, info, opt.linear_solver, ph, x

    dt, info, solver, ph, x = run(cfg.ordering)
    pat = solver.pattern
    fac = ph.get("chol_factor_levels") or ph.get("chol_factor_hblocks") or {"avg_ms": float("nan")}
    sol_ms = sum(v["total_ms"] for k_, v in ph.items() if k_.startswith("chol_solve")) / max(info.iters_done, 1)
    tf = B * pat.flops / (fac["avg_ms"] * 1e-3) / 1e12
    out = {
        "metric": "LM iterations/sec (batch x vars) on SE3 pose-graph", "value": B * info.iters_done / dt, "unit": "problem-iterations/s",
        "ms_per_step": dt / info.iters_done * 1e3, "steps": K_iters, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"SE3 pose-graph {P} poses / {len(edges)} edges (odometry chain + local loop closures, shuffled labels) + 1 "
                               f"prior, batch {B}, LM damping {cfg.damping} + HipSparseCholeskySolver(ordering={cfg.ordering!r})",
                   "poses": P, "edges": len(edges), "batch": B, "n": 6 * P,
                   "regime": "evaluations/pose_graph_synthetic.sh:5-12 (the reference's sparse-solver sweep: batch 8-256, to 4096 poses)"},
        "ordering": dict(solver.ordering_info, candidates=None), "levels": getattr(pat, "tree_levels", getattr(pat, "nlevels", pat.ntiles)), "launch_levels": getattr(pat, "nlevels", pat.ntiles),
        "subtree_streams": bool(getattr(pat, "two_streams", False)), "tiles": pat.ntiles,
        "tiles_of_L": pat.l_tiles, "level_scheduled": bool(solver.levels),
        "factor_ms": fac["avg_ms"], "solves_ms_per_iteration": sol_ms, "executed_GFLOP_per_factorisation": pat.flops / 1e9,
        "executed_TFLOPs": tf, "executed_frac_of_peak": tf / PEAK["f32"],
        "roofline": {"bound": "mfma", "kernel": "thx_chol_factor_levels (one chol_diag | chol_syrk + chol_potrf launch and one chol_offdiag "
                                                "launch per elimination-tree level)", "achieved": tf, "peak": PEAK["f32"], "unit": "TFLOP/s",
                     "frac": tf / PEAK["f32"], "flops_per_launch": B * pat.flops, "avg_launch_ms": fac["avg_ms"], "traffic": None},
        "phases_ms_per_call": {k_: round(v["avg_ms"], 4) for k_, v in ph.items()},
        "mean_error": [float(info.err_history[:, 0].mean()), float(info.err_history[:, info.iters_done].mean())]}
    if cfg.compare_rcm and cfg.ordering != "rcm":
        dt2, info2, solver2, ph2, x2 = run("rcm")
        fac2 = ph2.get("chol_factor_hblocks") or {"avg_ms": float("nan")}
        out["column_by_column_rcm"] = {"ms_per_step": dt2 / info2.iters_done * 1e3, "value": B * info2.iters_done / dt2,
                                       "factor_ms": fac2["avg_ms"], "levels": solver2.pattern.ntiles,
                                       "executed_GFLOP_per_factorisation": solver2.pattern.flops / 1e9}
        out["speedup_vs_column_by_column"] = dt2 / dt
        out["parity"] = {"against": "the same inputs through the column-by-column schedule under reverse Cuthill-McKee (another "
                                    "ordering, another factor); the level schedule against the REFERENCE: tests/test_gpu_full_size.py "
                                    "(pg_full_f64_lm fixture, ordering='nd') and tests/test_gpu_sparse.py",
                         "max_abs_pose_diff": float((x - x2).abs().max()),
                         "rel_diff_final_cost": float(((info.err_history[:, -1] - info2.err_history[:, -1]).abs() / info2.err_history[:, -1]).max())}
    del x, poses0, gt, meas, start
    free_device_memory()
    return out


def small_batch_run(cfg, ctx):
    """The headline graph (256 poses / 1024 edges, dense Cholesky) at the reference's own batch sizes (evaluations/
    pose_graph_synthetic.sh:7: 8 - 256): how far below the batch-4096 headline the dense path runs when the batch does not fill
    the chip.  One point per batch size: LM problem-iterations/s, ms per iteration, the factorisation's fraction of the MFMA peak."""
    import theseus_amd as th
    from theseus_amd.utils import synthetic as syn
    P, E, K_iters, dtype, dev = cfg.poses, cfg.edges, cfg.steps, torch.float32, ctx.device
    n = 6 * P
    edges = syn.pose_graph_topology(P, E, topology_seed=0)
    points = {}
    for B in cfg.batches:
        obj = syn.build_pose_graph_objective(edges, P, dtype=dtype, device=dev)
        opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipCholeskySolver, max_iterations=K_iters, abs_err_tolerance=0.0,
                                    rel_err_tolerance=0.0, step_size=1.0)
        layer = th.TheseusLayer(opt)
        timer = KernelTimer(opt.linear_solver.K)
        inputs = syn.input_dict(syn.make_pose_graph_tensors(edges, P, B, dtype=dtype, device=dev, seed=77 + B))
        okw = dict(damping=cfg.damping, track_err_history=True)
        with torch.no_grad():
            opt.set_params(max_iterations=2)
            layer.forward(inputs, optimizer_kwargs=okw)
            opt.set_params(max_iterations=K_iters)
            torch.cuda.synchronize()
            timer.enabled = True
            t0 = time.perf_counter()
            sol, info = layer.forward(inputs, optimizer_kwargs=okw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            timer.enabled = False
        ph = timer.summary()
        for n_ in timer.events:
            setattr(opt.linear_solver.K, n_, getattr(type(opt.linear_solver.K), n_).__get__(opt.linear_solver.K))
        fac = ph.get("chol_factor_hblocks") or ph.get("chol_factor") or {"avg_ms": float("nan")}
        tf = B * n ** 3 / 3.0 / (fac["avg_ms"] * 1e-3) / 1e12
        rl_max = int(opt.linear_solver.K.chol_schedule.right_looking_max_batch)
        rl_max = 64 if rl_max < 0 else rl_max      # (include/theseus_hip.h: thx_chol_schedule.right_looking_max_batch; default at 12 block columns, fp32: 64)
        points[f"b{B}"] = {"value": B * info.iters_done / dt, "ms_per_step": dt / info.iters_done * 1e3, "factor_ms": fac["avg_ms"],
                           "factor_TFLOPs": tf, "factor_frac_of_peak": tf / PEAK["f32"],
                           "schedule": "right-looking (one workgroup per tile product)" if B <= rl_max else "left-looking",
                           "mean_error": [float(info.err_history[:, 0].mean()), float(info.err_history[:, info.iters_done].mean())]}
        if B <= rl_max:   # the same inputs through the left-looking schedule (what rounds 1-5 ran at every batch size)
            prev = opt.linear_solver.K.chol_right_looking_max_batch(0)
            try:
                with torch.no_grad():
                    layer.forward(inputs, optimizer_kwargs=okw)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _, info2 = layer.forward(inputs, optimizer_kwargs=okw)
                    torch.cuda.synchronize()
                    dt2 = time.perf_counter() - t0
            finally:
                opt.linear_solver.K.chol_right_looking_max_batch(prev)
            points[f"b{B}"]["left_looking"] = {"ms_per_step": dt2 / info2.iters_done * 1e3, "value": B * info2.iters_done / dt2,
                                               "final_mean_error": float(info2.err_history[:, info2.iters_done].mean())}
            points[f"b{B}"]["speedup_vs_left_looking"] = dt2 / info2.iters_done / (dt / info.iters_done)
            del info2
        del sol, info, inputs, layer, opt, obj
        free_device_memory()
    return {"metric": "LM iterations/sec (batch x vars) on SE3 pose-graph", "unit": "problem-iterations/s", "dtype": "f32", "steps": K_iters,
            "config": {"workload": f"SE3 pose-graph {P} poses / {E} edges + 1 prior, LM damping {cfg.damping} + dense Cholesky at batch "
                                   f"{list(cfg.batches)} (the reference's published batch range, evaluations/pose_graph_synthetic.sh:7)"},
            "points": points}


def simple_run(cfg, ctx):
    """BASELINE.json configs[0] -- examples/simple_example.py: fit y = v exp(x), ONE AutoDiffCostFunction on a 1-d Vector, batch 16,
    Gauss-Newton + dense Cholesky, implicit backward w.r.t. the abscissae -- on theseus_amd's own API (theseus_amd/euclidean.py:
    thx_block_assemble + the tiled Cholesky).  The plumbing case: what it reports is correctness.  The problem is linear in v, so
    the exact solution is v* = sum(y e^x) / sum(e^2x) and the gradient of a loss of v* follows by plain autograd of that
    closed form -- the parity check needs no oracle."""
    import theseus_amd as th
    dev, dt = ctx.device, torch.float64
    B, N = cfg.batch, cfg.points
    gen = torch.Generator().manual_seed(0)
    xs = torch.linspace(-1, 1, N, dtype=dt).view(1, -1).repeat(B, 1)
    amp = 0.5 + 0.2 * torch.rand(B, 1, dtype=dt, generator=gen)
    ys = (amp * xs.exp()).to(dev)
    x0 = (xs + 0.05 * torch.randn(B, N, dtype=dt, generator=gen)).to(dev)
    x, y, v = th.Variable(x0.clone(), name="x"), th.Variable(ys, name="y"), th.Vector(tensor=torch.ones(B, 1, dtype=dt, device=dev), name="v")

    def residual(optim_vars, aux_vars):
        return aux_vars[1].tensor - optim_vars[0].tensor * aux_vars[0].tensor.exp()
    obj = th.Objective(dtype=dt)
    obj.add(th.AutoDiffCostFunction([v], residual, N, aux_vars=[x, y], cost_weight=th.ScaleCostWeight(torch.ones(1, 1, dtype=dt, device=dev))))
    opt = th.GaussNewton(obj, max_iterations=10, linearization_kwargs=dict(kernels=ctx.kernels) if ctx.kernels is not None else None)
    layer = th.TheseusLayer(opt)

    def once():
        phi = x0.clone().requires_grad_(True)
        sol, info = layer.forward({"x": phi, "v": torch.ones(B, 1, dtype=dt, device=dev)}, optimizer_kwargs={"backward_mode": "implicit"})
        loss = ((sol["v"] - 0.5) ** 2).mean()
        loss.backward()
        return sol["v"].detach(), phi.grad, info
    once()                                                   # warm-up (kernel modules, buffers)
    if ctx.on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(cfg.steps):
        vsol, grad, info = once()
    if ctx.on_gpu:
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / cfg.steps
    ph = x0.clone().requires_grad_(True)
    vstar = (ys * ph.exp()).sum(1, keepdim=True) / (2 * ph).exp().sum(1, keepdim=True)
    ((vstar - 0.5) ** 2).mean().backward()
    iters = int(info.iters_done)
    return {"metric": "forward (Gauss-Newton) + implicit backward calls/s", "value": 1e3 / ms, "unit": "calls/s", "ms_per_step": ms,
            "dtype": "f64", "data": "synthetic" if ctx.on_gpu else "TEST-STANDIN", "iters_done": iters,
            "config": {"workload": f"examples/simple_example.py shape: {N}-point fit of y = v exp(x), AutoDiffCostFunction on Vector(1), "
                                   f"batch {B}, GaussNewton(max_iterations=10) + implicit backward"},
            "parity": {"against": "closed form v* = sum(y e^x) / sum(e^2x) and its autograd gradient (fp64)",
                       "max_abs_v_err": float((vsol - vstar.detach()).abs().max()),
                       "grad_x_rel_err": float((grad - ph.grad).abs().max() / ph.grad.abs().max())}}


def ba_run(cfg, ctx):
    """BASELINE.json configs[3]: bundle adjustment, 512 SE3 cameras / 8192 Point3 / 32768 robust Reprojection costs, batch 256,
    adaptive ellipsoidal LM; the reduced camera system (Schur complement, 3072 x 3072) goes through the tile-sparse MFMA
    Cholesky along its band.  One GPU, rank 0."""
    import numpy as np
    import theseus_amd as th
    from theseus_amd.utils.synthetic_ba import make_ba_objective
    C, Np, B, K_iters, W = cfg.cams, cfg.points, cfg.batch, cfg.steps, cfg.warmup
    dtype = torch.float32 if cfg.dtype == "f32" else torch.float64
    obj, meta = make_ba_objective(C, Np, B, dtype=dtype, device=ctx.device)
    opt = th.LevenbergMarquardt(obj, max_iterations=K_iters, abs_err_tolerance=0.0, rel_err_tolerance=0.0)
    layer = th.TheseusLayer(opt)
    solver = opt.linear_solver
    timer = KernelTimer(solver.K)
    kw = dict(damping=1e-2, adaptive_damping=True, ellipsoidal_damping=True)
    packed = solver.linearization.packed
    with torch.no_grad():
        if W > 0:
            packed.sync(deep=True)
            start = packed.clone_state()
            opt.set_params(max_iterations=W)
            layer.forward(None, optimizer_kwargs=dict(track_err_history=True, **kw))
            # the timed run starts from the same initial state as the warm-up did: the packed state buffers are put back
            # (outside the timed region; nothing else of the objective changed, so forward() does not re-pack 42 k variables)
            packed.swap_state(start, repoint=True)
        opt.set_params(max_iterations=K_iters)
        torch.cuda.synchronize()
        fv0 = solver.factor_version
        timer.enabled = True
        t0 = time.perf_counter()
        sol, info = layer.forward(None, optimizer_kwargs=dict(track_err_history=True, **kw))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timer.enabled = False
    solves = solver.factor_version - fv0
    iters = max(info.iters_done, 1)
    phases = timer.summary()
    pat = getattr(solver, "pattern", None) if getattr(solver, "sparse", False) else None
    nc = 6 * meta["num_cams"]
    levels = bool(getattr(solver, "levels", False))
    fname = "chol_factor_levels" if levels else ("chol_factor_sparse" if pat is not None else "chol_factor")
    fac = phases.get(fname, {"avg_ms": float("nan")})
    ordering_info = dict(getattr(solver, "ordering_info", {}) or {})
    ordering_info.pop("candidates", None)
    peak = PEAK[cfg.dtype]
    dense_flops = B * nc ** 3 / 3.0
    executed = B * pat.flops if pat is not None else dense_flops
    kernel_ms = sum(v["total_ms"] for v in phases.values()) / max(solves, 1)
    ba_traffic = {}
    try:  # HBM bytes of the factorisation / of one whole linear solve, measured offline with rocprofv3 --pmc (profiles/traffic.json)
        ba_traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(
            f"ba_{cfg.dtype}_c{meta['num_cams']}_o{meta['num_obs']}_b{B}", {})
    except (OSError, ValueError):
        pass
    result = {
        "metric": "LM iterations/sec (batch x vars) on bundle adjustment", "value": B * iters / dt,
        "unit": "problem-iterations/s", "ms_per_step": dt / iters * 1e3, "dtype": cfg.dtype, "steps": K_iters, "warmup": W,
        "iters_done": info.iters_done, "linear_solves": solves,
        "config": {"workload": f"bundle adjustment {meta['num_cams']} SE3 cameras / {Np} Point3 ({meta['num_points']} observed) / "
                               f"{meta['num_obs']} Huber-robust Reprojection costs + Difference regularisers, batch {B}, adaptive "
                               f"ellipsoidal LM (damping 1e-2) + Schur complement + tile-sparse Cholesky of the {nc} x {nc} "
                               f"reduced camera system",
                   "cameras": meta["num_cams"], "points": meta["num_points"], "observations": meta["num_obs"], "batch": B,
                   "n": meta["n"], "n_reduced": nc},
        "mean_error": [float(info.err_history[:, 0].mean()), float(info.err_history[:, info.iters_done].mean())],
        "roofline": {
            "bound": "mfma", "kernel": f"thx_{fname} on the reduced camera system (chol_diag + chol_offdiag launches per "
                                       + ("elimination-tree level of the cameras' tile-level nested dissection, S read from the "
                                          "block list thx_ba_schur_blocks wrote" if levels else "block column") +
                                       ", non-zero tiles only)",
            # EXECUTED flops (the band of the reduced system: structurally zero tiles are skipped) / HIP-event time
            "achieved": executed / (fac["avg_ms"] * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
            "frac": executed / (fac["avg_ms"] * 1e-3) / 1e12 / peak, "traffic": ba_traffic.get("bytes_per_factor_call"),
            "traffic_unit": "bytes per factor call (PMC, rocprofv3)", "traffic_source": ba_traffic.get("source"),
            "traffic_per_solve": ba_traffic.get("bytes_per_solve"),
            "flops_per_launch": executed, "avg_launch_ms": fac["avg_ms"],
            "executed": None if pat is None else {
                "of_dense": pat.flops / pat.dense_flops, "tiles_of_L": [pat.l_tiles, pat.ntiles * (pat.ntiles + 1) // 2],
                "dense_equivalent_TFLOPs": dense_flops / (fac["avg_ms"] * 1e-3) / 1e12},
            # the MFMA floor of one linear solve on executed flops, against the measured time per linear solve
            "iteration": {"floor_ms": {"mfma_executed": round(executed / (peak * 1e12) * 1e3, 3)},
                          "kernel_ms_per_solve": round(kernel_ms, 3), "ms_per_solve": dt / max(solves, 1) * 1e3,
                          "device_gap_frac": round(1.0 - kernel_ms / (dt / max(solves, 1) * 1e3), 4),
                          "frac": executed / (peak * 1e12) * 1e3 / (dt / max(solves, 1) * 1e3)}},
        "phases_ms_per_call": {k: round(v["avg_ms"], 4) for k, v in phases.items()},
        "reduced_system": {"level_scheduled": levels, "ordering": ordering_info,
                           "levels": int(pat.tree_levels) if levels else None, "launch_levels": int(pat.nlevels) if levels else None,
                           "subtree_streams": bool(getattr(pat, "two_streams", False)), "tiles": int(pat.ntiles) if pat is not None else None,
                           "S": "block list (36 contiguous values per camera-pair block)" if levels else "dense frame"},
    }
    del sol, info, layer, opt, obj, timer, solver, packed
    free_device_memory()
    # ---- parity: the HIP path in fp64 at THIS size against the REAL reference's dense run (tests/golden/ba_full_f64_lm.npz:
    #      512 cameras / 8192 points / 32768 observations, one problem, two adaptive LM iterations; oracle/gen_golden.py) ----
    if cfg.parity:
        try:
            from tests.ba_common import run_ba
            from tests.helpers import load_golden
            g = load_golden("ba_full_f64_lm")
            cams, pts, used, _, pinfo, _ = run_ba(th, g, None, str(ctx.device))
            k = min(pinfo.err_history.shape[1], g["err_history"].shape[1])
            result["parity"] = {
                "reference": "the reference's DenseLinearization + CholeskyDenseSolver run at 512 / 8192 / 32768, fp64, one "
                             "problem, 2 adaptive LM iterations (tests/golden/ba_full_f64_lm.npz); HIP path in fp64",
                "hip_max_abs_camera_err": float(np.abs(cams.cpu().numpy() - g["final_cams"]).max()),
                "hip_max_abs_point_err": float(np.abs(pts.cpu().numpy() - g["final_pts"][:, used]).max()),
                "hip_rel_err_cost_history": float(np.abs(pinfo.err_history[:, :k].numpy() / g["err_history"][:, :k] - 1).max())}
            del cams, pts, pinfo
        except FileNotFoundError as e:
            result["parity"] = {"error": f"fixture missing: {e}"}
        free_device_memory()
    # ---- CPU baseline: the dense oracle (oracle/ba.py + oracle.pose_graph.lm_optimize: DenseLinearization + CholeskyDenseSolver
    #      restated) at the 32-CAMERA size of tests/golden/ba_mid_f64_lm.npz -- at 512 cameras the dense A is 20.6 GB per
    #      problem (one reference iteration there: ~5 min on 6 cores), not a bounded sample ----
    if cfg.cpu_baseline:
        try:
            from oracle import pose_graph as opg
            from tests.helpers import ba_problem, load_golden
            g = load_golden("ba_mid_f64_lm")
            p, state0, kwm, _ = ba_problem(g)
            kwm = dict(kwm)
            t0 = time.perf_counter()
            with torch.no_grad():
                _, oinfo = opg.lm_optimize(p, state0, abs_err_tolerance=0.0, rel_err_tolerance=0.0, **kwm)
            cpu_s = time.perf_counter() - t0
            Bm = state0[0].shape[0]
            result["cpu_baseline"] = {
                "value": Bm * max(oinfo.iters_done, 1) / cpu_s, "unit": "problem-iterations/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"REDUCED SIZE: {int(g['C'])} cameras / {int(g['Np'])} points / {g['obs_cam'].shape[0]} observations, "
                          f"{Bm} problems x {oinfo.iters_done} LM iterations ({cpu_s:.1f} s), dense oracle (n = "
                          f"{6 * int(g['C'])} + 3 x points); the 512-camera dense formulation is 20.6 GB of A per problem",
                # NOT measured in this run (it takes 9 minutes and 30 GB): the UNMODIFIED reference at THIS leg's size, recorded
                # when tests/golden/ba_full_f64_lm.npz was (re)generated -- profiles/r4/d_reference_ba_full_size_cpu_timing.txt
                "reference_full_size": {
                    "value": 2 / 532.0, "unit": "problem-iterations/s", "cores": 8, "kind": "reference", "dtype": "f64",
                    "where": "build container (not this box), oracle/gen_golden.py ba_full_f64_lm",
                    "sample": "512 cameras / 8192 points / 32768 observations, ONE problem x 2 adaptive LM iterations = 532.0 s "
                              "(theseus LevenbergMarquardt + DenseLinearization + CholeskyDenseSolver, dense A 20.6 GB)",
                    "speedup_of_this_leg": result["value"] / (2 / 532.0)}}
        except FileNotFoundError as e:
            result["cpu_baseline"] = {"error": f"fixture missing: {e}"}
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="problems per GPU (weak scaling), or per sub-batch")
    ap.add_argument("--total-batch", type=int, default=0,
                    help="STRONG scaling (BASELINE.json configs[2]: 32768 problems sharded over the GPUs): total problems of "
                         "the job; each rank solves its share in sub-batches of at most --batch problems, one after the "
                         "other, with persistent workspaces.  0 (default) = weak scaling, --batch problems per GPU")
    ap.add_argument("--poses", type=int, default=256)
    ap.add_argument("--edges", type=int, default=1024)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--damping", type=float, default=1e-3)
    ap.add_argument("--adaptive", action="store_true")
    ap.add_argument("--solver", default="dense", choices=["dense", "sparse"],
                    help="dense: HipCholeskySolver (n^3/3 per problem); sparse: HipSparseCholeskySolver -- the same tile kernels "
                         "under a reverse Cuthill-McKee variable ordering, skipping the tiles of L (and the K-loop blocks) that "
                         "are structurally zero")
    ap.add_argument("--cpu-sample", type=int, default=128, help="problems in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-chunk", type=int, default=32, help="problems per CPU-baseline chunk")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--parity-sample", type=int, default=32, help="problems in the exact-parity sub-sample (0 = skip)")
    ap.add_argument("--no-sparse-leg", action="store_true",
                    help="skip the second measurement of the same workload with the tile-sparse solver (reported under "
                         "'tile_sparse' next to the dense headline)")
    ap.add_argument("--implicit", action="store_true",
                    help="BASELINE.json configs[4]: forward LM + implicit backward (one backward linear solve) through "
                         "TheseusLayer; measurement tensors require grad; a step = one LM iteration of the forward")
    ap.add_argument("--legs", default="auto",
                    help="the other BASELINE configs under 'configs' in the same line: 'auto' (default: all of them when the "
                         "headline configuration is run unmodified), 'none', or a comma list of fp64,ba,implicit,strong")
    # TEST SEAM (tests/test_bench_cli.py): run the launcher / sharding / reporting logic without a GPU.  The numbers of such
    # a run are not measurements: the line says "data": "TEST-STANDIN" and carries no roofline.
    ap.add_argument("--test-kernels", default="", help=argparse.SUPPRESS)   # "module:Class" of a stand-in kernels class
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)
    ap.add_argument("--strong-total", type=int, default=32768, help=argparse.SUPPRESS)   # (the strong leg's job size; tests)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        respawn_under_torchrun(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    standin = bool(args.test_kernels)
    on_gpu = not standin
    # (THX_BENCH_ONE_DEVICE=1: every rank on cuda:0 -- a pre-flight of the sharded path with the real kernels on a 1-GPU box,
    #  over gloo; the driver's multi-GPU runs use one device per rank and RCCL)
    one_device = os.environ.get("THX_BENCH_ONE_DEVICE") == "1"
    device = torch.device("cuda", 0 if one_device else local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if on_gpu and args.backend == "nccl":
            dist.init_process_group(args.backend, device_id=device)
        else:
            dist.init_process_group(args.backend)
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    kernels = None
    if standin:
        mod, cls = args.test_kernels.split(":")
        kernels = getattr(importlib.import_module(mod), cls)()
    ctx = SimpleNamespace(world=world, rank=rank, device=device, on_gpu=on_gpu, kernels=kernels, dist=dist)
    head = SimpleNamespace(poses=args.poses, edges=args.edges, batch=args.batch, total_batch=args.total_batch, steps=args.steps,
                           warmup=args.warmup, dtype=args.dtype, damping=args.damping, adaptive=args.adaptive,
                           solver=args.solver, implicit=args.implicit, cpu_sample=args.cpu_sample, cpu_chunk=args.cpu_chunk,
                           cpu_iters=args.cpu_iters, parity_sample=args.parity_sample, sparse_leg=not args.no_sparse_leg)
    result = pg_run(head, ctx)

    # ---- the other BASELINE configs, as legs of the same line ----
    unmodified = (args.dtype == "f32" and args.solver == "dense" and not args.implicit and args.total_batch == 0
                  and not args.adaptive and args.poses == 256 and args.edges == 1024 and args.batch == 4096)
    if args.legs == "auto":
        legs = (["fp64", "ba", "implicit", "simple", "sparse", "small"] if world == 1 else ["strong"]) if (unmodified and on_gpu) else []
    elif args.legs == "none":
        legs = []
    else:
        legs = [x for x in args.legs.split(",") if x]
    configs = {}

    def leg(name, fn):
        t0 = time.perf_counter()
        try:
            r = fn()
        except Exception as e:   # a leg must not take the headline down with it; the line says what happened
            import traceback
            traceback.print_exc()
            if world > 1:
                raise            # (collectives in flight: the other ranks cannot be left waiting)
            r = {"error": f"{type(e).__name__}: {e}"}
            free_device_memory()
        if r is not None:
            r["leg_wall_s"] = round(time.perf_counter() - t0, 1)
            configs[name] = r

    def variant(**kw):
        d = dict(vars(head))
        d.update(kw)
        return SimpleNamespace(**d)

    if "fp64" in legs and world == 1:
        leg("fp64_b4096", lambda: pg_run(variant(dtype="f64", cpu_sample=min(args.cpu_sample, 32), sparse_leg=False), ctx))
    if "ba" in legs and world == 1 and not standin:
        leg("ba_512_8192_32768_b256", lambda: ba_run(SimpleNamespace(cams=512, points=8192, batch=256, dtype="f32", steps=10,
                                                                     warmup=2, parity=args.parity_sample > 0,
                                                                     cpu_baseline=args.cpu_sample > 0), ctx))
    if "implicit" in legs and world == 1:
        leg("implicit_b1024", lambda: pg_run(variant(implicit=True, batch=min(1024, args.batch), sparse_leg=False,
                                                     cpu_sample=min(args.cpu_sample, 4), parity_sample=min(args.parity_sample, 4)),
                                             ctx))
    if not os.environ.get("THX_REFERENCE_ROOT") and world == 1 and not standin and legs and ("dropin" in legs or args.legs == "auto"):
        configs["dropin_real_theseus_loop"] = {
            "skipped": "reference absent (no THX_REFERENCE_ROOT on this box): the REAL theseus loop over theseus_amd.plugin cannot run "
                       "here", "recorded": {"fraction_of_mirror_loop": [0.909, 0.949], "tests_on_cuda": "43 passed",
                                            "source": "profiles/r6/z_bench_dropin_leg.json, profiles/r6/z_pytest_plugin_cuda.txt "
                                                      "(tools/dropin_gpu.sh stages a copy of the reference for one gpurun call)"}}
    if os.environ.get("THX_REFERENCE_ROOT") and world == 1 and not standin and ("dropin" in legs or args.legs == "auto"):
        # the DROP-IN on hardware: the REAL theseus loop (its Objective / LevenbergMarquardt / TheseusLayer) with theseus_amd.plugin
        # behind it on the headline workload, next to theseus_amd's own loop on the same inputs (tools/dropin_bench.py; only where
        # a copy of the reference is importable: tools/dropin_gpu.sh stages one for a single gpurun call)
        def dropin():
            import subprocess
            out = {}
            for tag, extra in (("checked", []), ("lagged_failure_check", ["--lagged"])):
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dropin_bench.py"), "--steps", "10", *extra],
                                   capture_output=True, text=True, timeout=600)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                out[tag] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
            best = out["checked"]
            return {"metric": "LM iterations/sec through the REAL theseus loop + theseus_amd.plugin", "unit": "problem-iterations/s",
                    "value": best.get("dropin_problem_iterations_per_s"), "mirror_loop": best.get("mirror_problem_iterations_per_s"),
                    "fraction_of_mirror_loop": (best["dropin_problem_iterations_per_s"] / best["mirror_problem_iterations_per_s"]
                                                if "dropin_problem_iterations_per_s" in best else None), "runs": out}
        leg("dropin_real_theseus_loop", dropin)
    if "sparse" in legs and world == 1 and not standin:
        for B_ in (64, 256):
            leg(f"sparse_4096_b{B_}", lambda: sparse_run(SimpleNamespace(poses=4096, batch=B_, steps=10, warmup=2, damping=1e-2,
                                                                          ordering="auto", compare_rcm=True), ctx))
    if "small" in legs and world == 1 and not standin:
        leg("dense_small_batch", lambda: small_batch_run(SimpleNamespace(poses=256, edges=1024, batches=(8, 16, 64, 256), steps=10,
                                                                         damping=1e-3), ctx))
    if "simple" in legs and world == 1:
        leg("simple_example_b16", lambda: simple_run(SimpleNamespace(batch=16, points=20, steps=5), ctx))
    if "strong" in legs:
        # BASELINE.json configs[2]: 32768 fp64 problems over the N GPUs of the node, each rank's share in sub-batches of 4096
        leg("strong_f64_32768" if args.strong_total == 32768 else f"strong_{args.dtype if standin else 'f64'}_{args.strong_total}",
            lambda: pg_run(variant(dtype=args.dtype if standin else "f64", total_batch=args.strong_total,
                                   batch=min(4096, args.batch), sparse_leg=False), ctx))
    if rank == 0:
        if configs:
            result["configs"] = configs
        print(json.dumps(compact(result)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
