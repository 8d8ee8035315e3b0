"""Bundle adjustment through the host path on CPU (PackedBA, HipSchurLinearization / HipSchurSolver, the LM loop on a
(cameras, points) state) with the TEST stand-in kernels: reproduces the reference's DenseLinearization +
CholeskyDenseSolver trajectories.  GPU twin: tests/test_gpu_ba.py."""
import numpy as np
import pytest

from tests.ba_common import reference_columns, run_ba
from tests.helpers import load_golden


@pytest.mark.parametrize("name", ["ba_f64_lm", "ba_f64_gn"])
def test_ba_host_path_matches_reference(name):
    import theseus_amd as th
    from tests.oracle_kernels import OracleKernels
    g = load_golden(name)
    cams, pts, used, deltas, info, opt = run_ba(th, g, OracleKernels())
    np.testing.assert_allclose(cams.numpy(), g["final_cams"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(pts.numpy(), g["final_pts"][:, used], rtol=0, atol=1e-5)
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    np.testing.assert_allclose(info.err_history[:, :k].numpy(), g["err_history"][:, :k], rtol=1e-6)
    cols, _ = reference_columns(g)
    np.testing.assert_allclose(deltas[0].numpy()[:, cols], g["delta"][0], rtol=0, atol=1e-7 * max(1.0, np.abs(g["delta"][0]).max()))
    lin = opt.linear_solver.linearization
    assert lin.num_cols == int(g["num_cols"]) and lin.num_rows == int(g["num_rows"])
