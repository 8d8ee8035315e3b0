"""-m gpu, collected LAST (features added after the round's last GPU run): BackwardMode.UNROLL / TRUNCATED on the generic path through the HIP kernels (thx_block_assemble, the
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


def test_ba_with_camera_camera_costs_on_the_gpu():
    """Bundle adjustment with Between (odometry) costs on consecutive cameras through the HIP kernels -- thx_pg_assemble /
    thx_pg_error over the camera buffer joined with the Schur complement (theseus_amd/ba.py) -- against the REAL reference's run
    (tests/golden/ba_f64_camcam_lm.npz).  CPU twin with the stand-in kernels: tests/test_ba_host.py."""
    import numpy as np
    import theseus_amd as th
    from tests.ba_common import reference_columns, run_ba
    g = load_golden("ba_f64_camcam_lm")
    cams, pts, used, deltas, info, opt = run_ba(th, g, None, "cuda")
    assert len(opt.linear_solver.linearization.packed.cc_costs) == g["cc_edges"].shape[0]
    np.testing.assert_allclose(cams.cpu().numpy(), g["final_cams"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(pts.cpu().numpy(), g["final_pts"][:, used], rtol=0, atol=1e-5)
    k = min(info.err_history.shape[1], g["err_history"].shape[1])
    np.testing.assert_allclose(info.err_history[:, :k].numpy(), g["err_history"][:, :k], rtol=1e-6)
    cols, _ = reference_columns(g)
    np.testing.assert_allclose(deltas[0].cpu().numpy()[:, cols], g["delta"][0], rtol=0, atol=1e-7 * max(1.0, np.abs(g["delta"][0]).max()))


def test_ba_with_camera_camera_costs_implicit_backward_on_the_gpu():
    """... and its implicit backward: all gradient groups of tests/golden/ba_f64_camcam_implicit.npz incl. the odometry measurements
    and weights (thx_pg_vjp over the camera columns of the backward solve)."""
    import numpy as np
    import theseus_amd as th
    from tests.ba_common import run_ba_implicit
    g = load_golden("ba_f64_camcam_implicit")
    got = run_ba_implicit(th, g, None, "cuda")
    np.testing.assert_allclose(got["final_cams"], g["final_cams"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(got["final_pts"], g["final_pts"], rtol=0, atol=1e-6)
    assert abs(got["loss"] - float(g["loss"])) < 1e-5
    for k in ("log_radius", "feat", "focal", "k1", "k2", "w_obs", "gt_cams", "w_strong", "w_reg", "cc_meas", "w_cc"):
        want = g["grad_" + k]
        np.testing.assert_allclose(got["grad_" + k], want, rtol=0, atol=5e-6 * max(np.abs(want).max(), 1e-12), err_msg=k)
