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


@pytest.mark.parametrize("tag", ["gn_unroll", "lm_trunc", "lm_ellips_unroll"])
@pytest.mark.parametrize("fixture", ["pg2_f64_unrolled", "pg3_f64_unrolled"])
def test_differentiating_through_the_iterations_of_se2_and_so3_pose_graphs(fixture, tag):
    """The 3-dof groups (thx_pg2_unroll_vjp / thx_pgso3_unroll_vjp; SE2: plain autograd everywhere, SO3: torchlie's conventions) on
    the problems of the implicit fixtures, differentiated through the reference's iterations (oracle/gen_golden.py:
    gen_pg23_unrolled)."""
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    run_pg_unrolled(th, load_golden(fixture), tag, "cpu", OracleKernels())


def test_bundle_adjustment_refuses_unrolled_differentiation():
    """Bundle adjustment does not differentiate through its iterations on the fused path: a loud NotImplementedError, no autograd /
    CPU fallback."""
    import torch
    import theseus_amd as th
    from tests.ba_common import run_ba_implicit
    from tests.oracle_kernels import OracleKernels
    g = load_golden("ba_f64_implicit")

    class Unroll:
        """theseus_amd with TheseusLayer.forward forced to backward_mode='unroll'."""
        def __getattr__(self, k):
            return getattr(th, k)

        class TheseusLayer(th.TheseusLayer):
            def forward(self, inputs=None, optimizer_kwargs=None):
                return super().forward(inputs, optimizer_kwargs=dict(optimizer_kwargs or {}, backward_mode="unroll"))
    with pytest.raises(NotImplementedError, match="implicit"):
        run_ba_implicit(Unroll(), g, OracleKernels(), "cpu")
