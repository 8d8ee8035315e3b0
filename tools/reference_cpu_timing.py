"""The UNMODIFIED reference (theseus: LevenbergMarquardt + DenseLinearization + CholeskyDenseSolver, vectorize=True) timed on THIS
container's host cores at the headline size -- 256 SE3 poses / 1024 Between edges + the 1e-3 prior, fp32, LM damping 1e-3 --
on 256 problems in chunks of 64, next to the torch-CPU port (oracle.pose_graph.lm_optimize) on the same data.
Needs /root/reference: build container only.  usage: python tools/reference_cpu_timing.py [problems] [chunk] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import gen_golden as gg
from oracle import pose_graph as opg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 64
IT = int(sys.argv[3]) if len(sys.argv) > 3 else 3
th, lieF = gg.import_reference()
dtype = torch.float32
t_ref = t_port = 0.0
err = []
for c in range(0, N, CH):
    d = gg.full_size_data(lieF, min(CH, N - c), 500 + c, dtype)
    obj, poses = gg.build_reference_objective(th, d, dtype)
    opt = th.LevenbergMarquardt(obj, linear_solver_cls=th.CholeskyDenseSolver, vectorize=True, abs_err_tolerance=0.0,
                                rel_err_tolerance=0.0, max_iterations=IT, step_size=1.0)
    with torch.no_grad():
        t0 = time.perf_counter()
        info = opt.optimize(track_err_history=True, damping=1e-3)
        t_ref += time.perf_counter() - t0
    ref_final = torch.stack([p.tensor for p in poses], 1)
    prob = opg.PGProblem(num_poses=d["P"], edges=d["edges"], meas=d["meas"], w_between=d["w_between"], prior_idx=d["prior_idx"],
                         prior_target=d["prior_target"], w_prior=d["w_prior"])
    with torch.no_grad():
        t0 = time.perf_counter()
        final, oinfo = opg.lm_optimize(prob, d["poses"], max_iterations=IT, damping=1e-3, abs_err_tolerance=0.0, rel_err_tolerance=0.0)
        t_port += time.perf_counter() - t0
    err.append(float((final - ref_final).abs().max()))
    print(f"chunk {c // CH}: reference {t_ref:.1f} s, port {t_port:.1f} s so far; cost {float(info.err_history[:, 0].mean()):.1f} -> "
          f"{float(info.err_history[:, -1].mean()):.1f}; max |port - reference| pose {err[-1]:.2e}", flush=True)
print(f"REFERENCE (theseus, torch-CPU, {torch.get_num_threads()} threads): {N} problems x {IT} LM iterations in chunks of {CH}: "
      f"{t_ref:.1f} s = {N * IT / t_ref:.2f} problem-iterations/s")
print(f"PORT (oracle.pose_graph.lm_optimize, same data, same threads): {t_port:.1f} s = {N * IT / t_port:.2f} problem-iterations/s")
