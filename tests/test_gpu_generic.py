"""-m gpu: generic objectives on theseus_amd's own API through the HIP kernels (theseus_amd/euclidean.py: thx_block_assemble,
the tiled Cholesky, thx_vec_retract / thx_copy_where / thx_lm_accept) -- BASELINE.json configs[0] (examples/simple_example.py)
and its two-variable LM variant against the REAL reference's run (tests/golden/simple_example.npz)."""
import pytest

from tests.helpers import load_golden
from tests.simple_example_common import check_simple_example, run_simple_example

pytestmark = pytest.mark.gpu


def test_simple_example_config0_matches_the_reference_on_the_gpu():
    import theseus_amd as th
    g = load_golden("simple_example")
    r = run_simple_example(th, g, "cuda")
    lin = r["opt"].linear_solver.linearization
    assert type(lin.packed).__name__ == "PackedEuclidean" and type(lin.K).__name__ == "HipKernels"
    check_simple_example(g, r)
