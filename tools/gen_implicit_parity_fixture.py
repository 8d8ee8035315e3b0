"""Writes tests/golden/bench_implicit_exact.npz: the EXACT evaluation (fp64 oracle) of bench.py's implicit leg on a 4-problem
sample of its workload -- 256 poses / 1024 edges (theseus_amd.utils.synthetic, seed 4321, generated on the CPU with the oracle's
SE3 ops), forward LM (cpu_iters - 1 iterations) + the grad-enabled Gauss-Newton step + backward of the gauge-free loss
sum(X_k^-1 X_{k+1}) -- so that the leg does not spend ~150 s of every bench run re-deriving numbers that never change:
  ex_final / ex_grad      fp64 arithmetic with the fp32 Taylor thresholds on the fp32 inputs (what an fp32 run approximates),
  ex64_final / ex64_grad  plain fp64 on the inputs projected onto the manifold (the reference of the leg's fp64 re-run).
The oracle is pinned to the reference's implicit gradients by tests/test_oracle_golden.py.  ~3 min on 8 cores.
usage: python tools/gen_implicit_parity_fixture.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from tests.oracle_kernels import OracleKernels  # noqa: E402
from theseus_amd.utils import synthetic as syn  # noqa: E402

P, E, SP, CI, DAMPING, SEED = 256, 1024, 4, 3, 1e-3, 4321


def main():
    edges = syn.pose_graph_topology(P, E, topology_seed=0)
    tensors = syn.make_pose_graph_tensors(edges, P, SP, dtype=torch.float32, device="cpu", seed=SEED, kernels=OracleKernels())
    inputs = syn.input_dict(tensors)
    with bench.limited_threads(8):
        ex_final, ex_grad, s1 = bench.oracle_implicit(tensors, edges, P, torch.float32, SP, CI, DAMPING, exact=True)
        t64 = {k: bench.on_manifold(t) for k, t in tensors.items() if t.dim() == 3 and t.shape[-2:] == (3, 4)}
        ex64_final, ex64_grad, s2 = bench.oracle_implicit(t64, edges, P, torch.float64, SP, CI, DAMPING, exact=True)
    out = os.path.join(ROOT, "tests", "golden", "bench_implicit_exact.npz")
    np.savez_compressed(
        out, P=P, E=E, problems=SP, cpu_iters=CI, damping=DAMPING, seed=SEED, edges=np.asarray(edges),
        poses0=torch.stack([inputs[f"VERTEX_SE3__{k}"] for k in range(P)], 1).numpy(),
        meas=torch.stack([inputs[f"EDGE_SE3__{i}_{j}"] for (i, j) in edges], 1).numpy(),
        prior=inputs["VERTEX_SE3__0__PRIOR"].numpy(),
        ex_final=ex_final.numpy(), ex_grad=ex_grad.numpy(), ex64_final=ex64_final.numpy(), ex64_grad=ex64_grad.numpy())
    print(f"wrote {out} ({os.path.getsize(out) / 1e6:.2f} MB): exact runs {s1:.0f} s + {s2:.0f} s")


if __name__ == "__main__":
    main()
