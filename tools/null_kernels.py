"""Profiling aid (no GPU): the TEST stand-in kernels with the heavy ones turned into no-ops, so that what a profile of
tools/dropin_bench.py --test-kernels tools.null_kernels:NullKernels shows is the HOST side of the drop-in (the reference's loop +
the plugin's Python) and nothing else.  Outputs stay whatever the buffers held (zeros): the numbers are meaningless."""
from tests.oracle_kernels import OracleKernels


class NullKernels(OracleKernels):
    name = "null"

    def _nop(self, *a, **k):
        return None

    def pg_error(self, s, t, partials, err, poses=None):
        err.zero_()

    def retract(self, poses, delta, step, ignore_mask, out):
        out.copy_(poses)


for _n in ("pg_assemble", "pg_jacobians", "chol_factor", "chol_solve_backward", "chol_solve", "lm_accept", "lm_accept_diag", "diag"):
    setattr(NullKernels, _n, NullKernels._nop)
