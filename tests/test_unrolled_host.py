"""CPU twin (TEST stand-in kernels) of the fused unrolled backward of SE3 pose graphs: the host logic of
theseus_amd/autograd.py:PGUnrolledIteration -- the chain over iterations, the retraction's two paths, the factor copies, the fixed
order sum of the poses' gradients, accept / reject selects -- against the REAL reference's gradients.  The kernel's maths itself is
checked without a GPU by tests/test_unroll_math_host.py, the kernel on the GPU by tests/test_gpu_unrolled.py."""
import pytest

from tests.helpers import load_golden
from tests.unrolled_common import run_pg_unrolled


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_unroll", "lm_trunc", "lm_ellips_unroll", "gn_trunc_conv"])
def test_differentiating_through_the_iterations_of_a_pose_graph(tag):
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    run_pg_unrolled(th, load_golden("pg_f64_unrolled"), tag, "cpu", OracleKernels())


def test_robust_costs_are_refused_loudly():
    import torch
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = load_golden("pg_f64_unrolled")
    t = torch.from_numpy
    meas = t(g["meas"]).requires_grad_(True)
    obj = th.Objective(dtype=torch.float64)
    poses = [th.SE3(tensor=t(g["poses0"])[:, k].clone(), name=f"pose_{k}") for k in range(int(g["P"]))]
    radius = th.Vector(tensor=torch.zeros(1, 1, dtype=torch.float64), name="log_loss_radius")
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        cf = th.Between(poses[i], poses[j], th.SE3(tensor=meas[:, k], name=f"meas_{k}"),
                        th.DiagonalCostWeight(th.Variable(t(g["w_between"])[:, k], name=f"w_{k}")), name=f"between_{k}")
        obj.add(th.RobustCostFunction(cf, th.WelschLoss, radius, name=f"robust_{k}"))
    obj.add(th.Difference(poses[0], th.SE3(tensor=t(g["prior_target"])[:, 0], name="tgt"),
                          th.ScaleCostWeight(torch.tensor(1.0, dtype=torch.float64)), name="prior"))
    opt = th.LevenbergMarquardt(obj, max_iterations=2, linearization_kwargs=dict(kernels=OracleKernels()))
    with pytest.raises(NotImplementedError, match="without robust cost functions"):
        th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(backward_mode="unroll", damping=0.1))
