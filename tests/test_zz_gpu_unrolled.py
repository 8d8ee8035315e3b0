"""-m gpu, collected LAST: BackwardMode.UNROLL / TRUNCATED on the generic path through the HIP kernels (thx_block_assemble, the
tiled Cholesky's damped factorisation, thx_chol_solve with a copy of each iteration's factor in the backward) against the REAL
reference's gradients (tests/golden/simple_example.npz).  CPU twin with the stand-in kernels: tests/test_generic_host.py."""
import pytest

from tests.helpers import load_golden
from tests.simple_example_common import run_unrolled

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["gn_unroll", "gn_trunc", "lm_unroll", "lm_trunc", "gn_trunc_conv"])
def test_differentiating_through_the_iterations_on_the_gpu(tag):
    import theseus_amd as th
    run_unrolled(th, load_golden("simple_example"), tag, "cuda")
