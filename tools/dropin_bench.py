"""The DROP-IN measured: the REAL ``theseus`` (its own Objective / cost functions / LevenbergMarquardt loop / TheseusLayer)
with ``theseus_amd.plugin`` behind it (HipLinearization + HipCholeskySolver + the Objective hook set), on the headline
workload of bench.py (BASELINE.json configs[1]: 256 SE3 poses / 1024 Between edges + prior, batch 4096, fp32, LM damping
1e-3, tolerances 0).  Needs the reference importable: THX_REFERENCE_ROOT (a scratch copy on the GPU box, tools/dropin_gpu.sh).

usage: python tools/dropin_bench.py [--batch 4096] [--steps 10] [--dtype f32] [--no-hooks]
Prints ONE JSON line: problem-iterations/s of the drop-in, of theseus_amd's own loop on the same inputs, and their agreement.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("THX_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)
for p in (os.path.join(ROOT, "oracle", "stubs"), REF, REF + "/torchlie", REF + "/torchkin"):
    sys.path.append(p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--poses", type=int, default=256)
    ap.add_argument("--edges", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--adaptive", action="store_true")
    ap.add_argument("--lagged", action="store_true", help="linear_solver_kwargs=dict(lagged_failure_check=True): no host sync in solve()")
    ap.add_argument("--no-hooks", action="store_true", help="only Linearization + LinearSolver replaced (round-1 boundary)")
    ap.add_argument("--reference-gpu-batch", type=int, default=0,
                    help="also run the UNMODIFIED reference (DenseLinearization + CholeskyDenseSolver through PyTorch-ROCm) on the "
                         "same GPU at this batch size (its dense A is 37.8 MB per problem in fp32).  OFF by default: on this image "
                         "(torch 2.10.0+rocm7.0) its linear solve ends in 'HIP error: unspecified launch failure' at n = 1536 in "
                         "both dtypes (profiles/r2/r_dropin_bench_*_with_reference_on_gpu.json) -- do not put it in a routine run")
    ap.add_argument("--profile", action="store_true", help="cProfile of the drop-in's optimize() (host side), top functions to stderr")
    ap.add_argument("--test-kernels", default="", help=argparse.SUPPRESS)   # dry run of this script without a GPU
    args = ap.parse_args()
    import warnings
    warnings.filterwarnings("ignore")
    import theseus as th
    import theseus_amd as ta
    import theseus_amd.plugin as thp
    from theseus_amd.utils import synthetic as syn
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    kernels, sync = None, torch.cuda.synchronize
    dev = torch.device("cuda", 0)
    if args.test_kernels:
        import importlib
        mod, cls = args.test_kernels.split(":")
        kernels, dev, sync = getattr(importlib.import_module(mod), cls)(), torch.device("cpu"), (lambda: None)
    lkw = dict(kernels=kernels) if kernels is not None else {}
    P, E, B, K = args.poses, args.edges, args.batch, args.steps
    edges = syn.pose_graph_topology(P, E, topology_seed=0)
    tensors = syn.make_pose_graph_tensors(edges, P, B, dtype=dtype, device=dev, seed=1234, kernels=kernels)
    inputs = syn.input_dict(tensors)

    # ---- the reference's own objective, as examples/pose_graph/pose_graph_synthetic.py:130-152 builds it ----
    obj = th.Objective(dtype=dtype)
    eye = torch.eye(3, 4, dtype=dtype).view(1, 3, 4)
    poses = [th.SE3(tensor=eye.clone(), name=f"VERTEX_SE3__{k}") for k in range(P)]
    w = th.DiagonalCostWeight(th.Variable(torch.tensor([[1 / syn.TRANSLATION_NOISE] * 3 + [1 / syn.ROTATION_NOISE] * 3], dtype=dtype),
                                          name="EDGE_WEIGHT"))
    for (i, j) in edges:
        obj.add(th.Between(poses[i], poses[j], th.SE3(tensor=eye.clone(), name=f"EDGE_SE3__{i}_{j}"), w, name=f"between_{i}_{j}"))
    obj.add(th.Difference(poses[0], th.SE3(tensor=eye.clone(), name="VERTEX_SE3__0__PRIOR"),
                          th.ScaleCostWeight(th.Variable(torch.tensor([[syn.PRIOR_WEIGHT]], dtype=dtype), name="PRIOR_WEIGHT")),
                          name="pose_prior"))
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=thp.HipCholeskySolver, linearization_cls=thp.HipLinearization,
                                linearization_kwargs=dict(objective_hooks=not args.no_hooks, **lkw), max_iterations=K,
                                linear_solver_kwargs=dict(lagged_failure_check=args.lagged),
                                abs_err_tolerance=0.0, rel_err_tolerance=0.0, step_size=1.0)
    layer = th.TheseusLayer(opt)
    layer.to(dev)
    okw = dict(damping=1e-3, adaptive_damping=args.adaptive)

    def run(l, iters, inp=None):
        l.optimizer.set_params(max_iterations=iters)
        sync()
        t0 = time.perf_counter()
        with torch.no_grad():
            sol, info = l.forward(inputs if inp is None else inp, optimizer_kwargs=dict(track_err_history=True, **okw))
        sync()
        return sol, info, time.perf_counter() - t0

    run(layer, 2)
    if args.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        run(layer, K)
        pr.disable()
        st = pstats.Stats(pr, stream=sys.stderr)
        st.sort_stats("cumulative").print_stats(45)
        st.sort_stats("tottime").print_stats(25)
    sol, info, dt = run(layer, K)
    iters = int(info.err_history.shape[1] - 1)
    # ---- theseus_amd's own loop on the same inputs ----
    mobj = syn.build_pose_graph_objective(edges, P, dtype=dtype, device=dev)
    mopt = ta.LevenbergMarquardt(mobj, linear_solver_cls=ta.HipCholeskySolver, max_iterations=K, abs_err_tolerance=0.0,
                                 rel_err_tolerance=0.0, step_size=1.0, linearization_kwargs=lkw or None)
    mlayer = ta.TheseusLayer(mopt)
    mlayer.optimizer = mopt
    run(mlayer, 2)
    msol, minfo, mdt = run(mlayer, K)
    ref_gpu = None
    if args.reference_gpu_batch > 0:
        # ---- the reference as it is, on the same device: its own vectorised torch cost functions, dense A / AtA, and
        #      torch.linalg.cholesky / cholesky_solve of PyTorch-ROCm ----
        Br, Kr = args.reference_gpu_batch, 3
        del layer, opt, obj, mlayer, mopt, mobj   # their Hessian / factor frames (2 x 38.6 GB each at the headline size)
        import gc
        gc.collect()
        if dev.type == "cuda":
            torch.cuda.empty_cache()
        robj = th.Objective(dtype=dtype)
        rposes = [th.SE3(tensor=eye.clone(), name=f"VERTEX_SE3__{k}") for k in range(P)]
        rw = th.DiagonalCostWeight(th.Variable(torch.tensor([[1 / syn.TRANSLATION_NOISE] * 3 + [1 / syn.ROTATION_NOISE] * 3], dtype=dtype),
                                               name="EDGE_WEIGHT"))
        for (i, j) in edges:
            robj.add(th.Between(rposes[i], rposes[j], th.SE3(tensor=eye.clone(), name=f"EDGE_SE3__{i}_{j}"), rw, name=f"between_{i}_{j}"))
        robj.add(th.Difference(rposes[0], th.SE3(tensor=eye.clone(), name="VERTEX_SE3__0__PRIOR"),
                               th.ScaleCostWeight(th.Variable(torch.tensor([[syn.PRIOR_WEIGHT]], dtype=dtype), name="PRIOR_WEIGHT")),
                               name="pose_prior"))
        ropt = th.LevenbergMarquardt(robj, linear_solver_cls=th.CholeskyDenseSolver, vectorize=True, max_iterations=Kr,
                                     abs_err_tolerance=0.0, rel_err_tolerance=0.0, step_size=1.0)
        rlayer = th.TheseusLayer(ropt)
        rlayer.to(dev)
        rin = {k: (v[:Br].contiguous() if v.shape[0] == B else v) for k, v in inputs.items()}
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            run(rlayer, 1, rin)
            torch.cuda.reset_peak_memory_stats() if dev.type == "cuda" else None
            rsol, rinfo, rdt = run(rlayer, Kr, rin)
        ref_warn = sorted({str(w.message)[:400] for w in caught if "linear optimizer" in str(w.message)})
        if ref_warn:   # what torch.linalg.cholesky said, and how far from positive definite the reference's AtA is
            lin = ropt.linear_solver.linearization
            try:
                ev = torch.linalg.eigvalsh(lin.AtA[:4].double().cpu())
                ref_warn.append(f"eigenvalues of its AtA (first 4 problems, fp64 on the host): min {ev.min().item():.3e}, max {ev.max().item():.3e}")
            except Exception as e:   # noqa: BLE001
                ref_warn.append(f"(eigvalsh of its AtA failed: {e})")
        # iterations the reference actually completed: a failed linear solve (torch.linalg.cholesky raising on a matrix that
        # is not positive definite in this precision) ends its loop with status FAIL and leaves inf in err_history
        riters = int(torch.isfinite(rinfo.err_history).all(dim=0).sum()) - 1
        ref_gpu = {"batch": Br, "lm_iterations_requested": Kr, "lm_iterations": riters,
                   "status": sorted({str(x).split(".")[-1] for x in rinfo.status.tolist()}), "warnings": ref_warn,
                   "problem_iterations_per_s": (Br * riters / rdt) if riters > 0 else None,
                   "ms_per_iteration": (rdt / riters * 1e3) if riters > 0 else None, "wall_ms": rdt * 1e3,
                   "peak_memory_GB": (torch.cuda.max_memory_allocated() / 1e9) if dev.type == "cuda" else None,
                   "mean_error": [float(rinfo.err_history[:, 0].mean()), float(rinfo.err_history[:, -1].mean())]}
        del rsol, rlayer, ropt, robj
    a = torch.stack([sol[f"VERTEX_SE3__{k}"] for k in range(P)], 1)
    b = torch.stack([msol[f"VERTEX_SE3__{k}"] for k in range(P)], 1)
    print(json.dumps({
        "what": "real theseus loop + theseus_amd.plugin vs theseus_amd's own loop, same inputs, same kernels",
        "hooks": not args.no_hooks, "lagged_failure_check": args.lagged, "dtype": args.dtype, "batch": B, "poses": P, "edges": E, "lm_iterations": iters,
        "dropin_problem_iterations_per_s": B * iters / dt, "dropin_ms_per_iteration": dt / iters * 1e3,
        "mirror_problem_iterations_per_s": B * minfo.iters_done / mdt, "mirror_ms_per_iteration": mdt / minfo.iters_done * 1e3,
        "max_abs_pose_difference": float((a - b).abs().max()),
        "dropin_mean_error": [float(info.err_history[:, 0].mean()), float(info.err_history[:, -1].mean())],
        "mirror_mean_error": [float(minfo.err_history[:, 0].mean()), float(minfo.err_history[:, minfo.iters_done].mean())],
        "reference_on_this_gpu": ref_gpu}))


if __name__ == "__main__":
    main()
