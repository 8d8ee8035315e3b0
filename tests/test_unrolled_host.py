"""CPU twin (TEST stand-in kernels) of the fused unrolled backward of SE3 pose graphs: the host logic of
theseus_amd/autograd.py:PGUnrolledIteration -- the chain over iterations, the retraction's two paths, the factor copies, the fixed
order sum of the poses' gradients, accept / reject selects -- against the REAL reference's gradients.  The kernel's maths itself is
checked without a GPU by tests/test_unroll_math_host.py, the kernel on the GPU by tests/test_gpu_unrolled.py."""
import pytest

from tests.helpers import load_golden
from tests.unrolled_common import run_pg_unrolled


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_unroll", "lm_trunc", "lm_ellips_unroll", "gn_trunc_conv", "lm_welsch_unroll",
                                 "gn_huberflat_trunc"])
def test_differentiating_through_the_iterations_of_a_pose_graph(tag):
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    run_pg_unrolled(th, load_golden("pg_f64_unrolled"), tag, "cpu", OracleKernels())


def test_other_groups_are_refused_loudly():
    """SE2 / SO3 pose graphs (and bundle adjustment) do not differentiate through their iterations on the fused path: a loud
    NotImplementedError, no autograd / CPU fallback (tests/test_generic_host.py::test_fused_path_refuses_unrolled_differentiation)."""
    import torch
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = load_golden("pg3_f64_implicit")
    t = torch.from_numpy
    meas = t(g["meas"]).clone().requires_grad_(True)
    obj = th.Objective(dtype=torch.float64)
    poses = [th.SO3(tensor=t(g["poses0"])[:, k].clone(), name=f"pose_{k}") for k in range(int(g["P"]))]
    for k in range(g["edges"].shape[0]):
        i, j = g["edges"][k].tolist()
        obj.add(th.Between(poses[i], poses[j], th.SO3(tensor=meas[:, k], name=f"meas_{k}"),
                           th.DiagonalCostWeight(th.Variable(t(g["w_between"])[:, k].clone(), name=f"w_{k}")), name=f"between_{k}"))
    opt = th.GaussNewton(obj, max_iterations=2, linearization_kwargs=dict(kernels=OracleKernels()))
    with pytest.raises(NotImplementedError, match="fused for SE3 pose graphs"):
        th.TheseusLayer(opt).forward(None, optimizer_kwargs=dict(backward_mode="unroll"))
