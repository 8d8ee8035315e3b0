"""EXPERIMENT (round 5): the headline LM loop as K independent part-batches, each a complete optimize() on its own HIP stream.

The problems of a batch are independent, and the iteration is one MFMA-bound phase (the factorisation, ~90 % of the step)
followed by HBM-bound ones (backward substitution, retraction, error, assembly).  Inside ONE stream they run one after the
other.  With the batch split into parts that are offset in time, the HBM-bound phases of one part can run underneath the
factorisation of another.  This script measures the upper bound of that before anything is built into the loop: K optimizers
over B / K problems each, one host thread + one stream per part, optionally started one linear solve apart.

usage: python tools/exp_pipeline_parts.py --parts 2 --batch 4096 --iters 20 [--offset 1] [--dtype f32]
(the factorisation's own two-stream schedule is switched by the environment: THX_CHOL_SPLIT_MIN=0 turns it off)
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import theseus_amd as th  # noqa: E402
from theseus_amd.utils import synthetic as syn  # noqa: E402


def make(P, E, B, dtype, seed, iters):
    edges = syn.pose_graph_topology(P, E, topology_seed=0)
    obj = syn.build_pose_graph_objective(edges, P, dtype=dtype, device="cuda:0")
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.HipCholeskySolver, max_iterations=iters, abs_err_tolerance=0.0,
                                rel_err_tolerance=0.0, step_size=1.0)
    layer = th.TheseusLayer(opt)
    tensors = syn.make_pose_graph_tensors(edges, P, B, dtype=dtype, device="cuda:0", seed=seed)
    return layer, opt, syn.input_dict(tensors)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--offset", type=int, default=0, help="1: part k starts when part k-1 has queued its first linear solve")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    sys.setswitchinterval(0.0005)
    dtype = torch.float32 if a.dtype == "f32" else torch.float64
    K = a.parts
    Bp = a.batch // K
    parts = [make(256, 1024, Bp, dtype, 1234 + k, a.iters) for k in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    okw = dict(damping=1e-3, track_err_history=True)

    def run(iters):
        for _, opt, _ in parts:
            opt.set_params(max_iterations=iters)
        evs = [torch.cuda.Event() for _ in range(K)]
        queued = [threading.Event() for _ in range(K)]
        infos = [None] * K
        errs = []

        def hook(k, opt):
            orig = opt.compute_delta
            state = {"n": 0}

            def wrapped(**kw):
                d = orig(**kw)
                if state["n"] == 0:
                    evs[k].record(torch.cuda.current_stream())
                    queued[k].set()
                state["n"] += 1
                return d
            opt.compute_delta = wrapped
            return orig

        def worker(k):
            layer, opt, inputs = parts[k]
            try:
                with torch.cuda.stream(streams[k]):
                    if a.offset and k > 0:
                        queued[k - 1].wait()
                        streams[k].wait_event(evs[k - 1])
                    orig = hook(k, opt)
                    try:
                        _, infos[k] = layer.forward(inputs, optimizer_kwargs=okw)
                    finally:
                        opt.compute_delta = orig
                        queued[k].set()
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
                queued[k].set()

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if K == 1:
            worker(0)
        else:
            ths = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if errs:
            raise errs[0]
        return dt, infos

    run(a.warmup)
    out = []
    for _ in range(a.rounds):
        dt, infos = run(a.iters)
        it = infos[0].iters_done
        out.append({"ms_per_step": dt / it * 1e3, "value": a.batch * it / dt,
                    "final_err": [float(i.err_history[:, it].mean()) for i in infos]})
    print(json.dumps({"parts": K, "batch": a.batch, "dtype": a.dtype, "offset": a.offset,
                      "split_min": os.environ.get("THX_CHOL_SPLIT_MIN", "default"), "runs": out}))


if __name__ == "__main__":
    main()
