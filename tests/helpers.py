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


def ba_problem(g):
    """Bundle-adjustment fixture -> (oracle BAProblem, (cams0, pts0), optimizer kwargs).  Points that no observation
    touches are not optimisation variables of the reference objective: they are dropped and the rest re-indexed."""
    from oracle import ba as oba
    t = torch.from_numpy
    kinds, idxs = g["var_kind"].tolist(), g["var_idx"].tolist()
    used = sorted(i for k, i in zip(kinds, idxs) if k == 1)
    remap = {old: new for new, old in enumerate(used)}
    var_order = [("cam", i) if k == 0 else ("pt", remap[i]) for k, i in zip(kinds, idxs)]
    cost_order = [(("obs", "cam_prior", "pt_prior", "cam_between")[k], int(i)) for k, i in zip(g["cost_kind"].tolist(), g["cost_idx"].tolist())]
    dtype = t(g["cams0"]).dtype
    robust = str(g["robust"]) or None
    p = oba.BAProblem(
        num_cams=int(g["C"]), num_points=len(used), obs_cam=t(g["obs_cam"]), obs_pt=torch.tensor([remap[i] for i in g["obs_pt"].tolist()]),
        feat=t(g["feat"]), w_obs=torch.ones(1, g["obs_cam"].shape[0], 2, dtype=dtype), focal=t(g["focal"]), k1=t(g["k1"]), k2=t(g["k2"]),
        cam_prior_idx=t(g["cam_prior_idx"]), cam_prior_target=t(g["cam_prior_target"]), w_cam_prior=t(g["w_cam_prior"]),
        pt_prior_idx=torch.tensor([remap[i] for i in g["pt_prior_idx"].tolist()]),
        pt_prior_target=torch.zeros(1, g["pt_prior_idx"].shape[0], 3, dtype=dtype), w_pt_prior=t(g["w_pt_prior"]),
        var_order=var_order, cost_order=cost_order, robust_obs=robust,
        log_radius_obs=torch.full((1, 1, 1), float(g["log_radius"]), dtype=dtype) if robust else None,
        cc_edges=t(g["cc_edges"]) if "cc_edges" in g else None, cc_meas=t(g["cc_meas"]) if "cc_edges" in g else None,
        w_cc=t(g["w_cc"]) if "cc_edges" in g else None)
    kw = ast.literal_eval(str(g["opt_kwargs"]))
    return p, (t(g["cams0"]), t(g["pts0"])[:, used]), kw, used
