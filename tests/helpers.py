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
    )
    kw = ast.literal_eval(str(g["opt_kwargs"]))
    return p, t(g["poses0"]), kw
