"""The reference's own known-answer test for this path -- tests/theseus_tests/test_pgo_benchmark.py:34-39 (four outer
losses of examples/pose_graph/pose_graph_synthetic.py at rel=abs=1e-10) -- restated on the committed inputs
(tests/golden/pgo_kat.npz, produced by the reference's generator: oracle/gen_golden.py:gen_pgo_kat)."""
import torch

from oracle import lie
from tests.helpers import load_golden


def kat():
    return load_golden("pgo_kat")


def batch_slices(g):
    bs = int(g["batch_size"])
    return [slice(k * bs, (k + 1) * bs) for k in range(4)]


def pose_loss(poses, gt):
    """pose_graph_synthetic.py:58-72: sum over poses and problems of |log(pose^-1 gt)|; poses/gt (B,P,3,4).
    torch ops on CPU tensors (autograd follows torchlie's conventions, oracle/lie.py)."""
    a, b = poses.reshape(-1, 3, 4), gt.reshape(-1, 3, 4)
    xi, _ = lie.se3_log_jlog_autograd(lie.se3_compose(lie.se3_inverse(a), b))
    return xi.norm(dim=1).sum()


def outer_loop(g, inner_solve):
    """pose_graph_synthetic.py:203-262: Adam(lr) on log_loss_radius, one step per batch.  ``inner_solve(sl,
    log_radius (1,1) tensor requiring grad) -> final poses (B,P,3,4)`` attached to log_radius.  Returns the losses."""
    t = torch.from_numpy
    param = torch.nn.Parameter(torch.tensor([[float(g["log_radius0"])]], dtype=torch.float64))
    opt = torch.optim.Adam([param], lr=float(g["lr"]))
    losses = []
    for sl in batch_slices(g):
        gt = t(g["gt"][sl])
        with torch.no_grad():
            ref = pose_loss(t(g["poses0"][sl]), gt)
        final = inner_solve(sl, param.clone())
        opt.zero_grad()
        loss = (pose_loss(final.cpu(), gt) - ref) / ref  # the outer loss is the caller's (torch, CPU); .cpu() is differentiable
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses
