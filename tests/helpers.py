"""Shared test helpers (tests may import the oracle; product code may not)."""
import ast
import os

import numpy as np
import torch

from oracle import pose_graph as opg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def golden_problem(g):
    """Golden fixture -> (oracle PGProblem, initial poses, optimizer kwargs)."""
    t = torch.from_numpy
    p = opg.PGProblem(
        num_poses=int(g["P"]), edges=t(g["edges"]), meas=t(g["meas"]), w_between=t(g["w_between"]),
        prior_idx=t(g["prior_idx"]), prior_target=t(g["prior_target"]), w_prior=t(g["w_prior"]),
        group=str(g["group"]) if "group" in g else "SE3",
    )
    kw = ast.literal_eval(str(g["opt_kwargs"]))
    return p, t(g["poses0"]), kw


def f32_truth_problem(p, poses0):
    """The fp32 problem widened to fp64 (same values): evaluating the oracle on it, with the fp32
    Taylor thresholds, gives the exact value of what the reference's fp32 path approximates."""
    import dataclasses
    p64 = dataclasses.replace(p, meas=p.meas.double(), w_between=p.w_between.double(),
                              prior_target=p.prior_target.double(), w_prior=p.w_prior.double())
    return p64, poses0.double()


class f32_thresholds:
    """Context manager: make the fp64 oracle switch Taylor branches where the fp32 reference does
    (torchlie/torchlie/global_params.py:44-58 keys the thresholds by dtype)."""

    def __enter__(self):
        from oracle import lie, lie_se2
        self._saved = [(m, dict(m.EPS[torch.float64])) for m in (lie, lie_se2)]
        for m in (lie, lie_se2):
            m.EPS[torch.float64] = {k: float(np.float32(v)) for k, v in m.EPS[torch.float32].items()}

    def __exit__(self, *a):
        for m, saved in self._saved:
            m.EPS[torch.float64] = saved
