"""pytest plugin (``-p thx_reference_injection``) for running the REFERENCE's own test files with theseus_amd's plugin classes
substituted for ``th.CholeskyDenseSolver`` / ``th.DenseLinearization`` ("the same files are the regression
suite to re-run with the new linearization_cls / linear_solver_cls injected").  Driven by tests/test_reference_suite_injected.py
with cwd = the reference checkout (its top-level ``tests`` package wins over this repo's, so the stand-in kernels are loaded by
FILE PATH under another module name).  The reference's tests build their objectives on the CPU, so the kernels behind the plugin
are the TEST stand-in (tests/oracle_kernels.py): this proves the INTERFACE; the kernels are proven by the -m gpu tests and by
tests/test_plugin_reference.py on a GPU box (tools/dropin_gpu.sh)."""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _standin_kernels():
    name = "thx_oracle_kernels"
    if name not in sys.modules:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "tests", "oracle_kernels.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    return sys.modules[name].OracleKernels()


def pytest_configure(config):
    if REPO not in sys.path:
        sys.path.append(REPO)          # appended: the reference's own ``tests`` package must keep its name
    import warnings
    warnings.filterwarnings("ignore")
    import theseus as th
    import theseus.optimizer as tho
    import theseus.optimizer.linear as thl
    import theseus.optimizer.nonlinear.nonlinear_least_squares as nls
    import theseus_amd.plugin as thp
    ref_dense = tho.DenseLinearization

    class InjectedLinearization(thp.HipLinearization):
        def __init__(self, objective, ordering=None, **kwargs):
            if kwargs.get("kernels") is None:
                kwargs["kernels"] = _standin_kernels()
            super().__init__(objective, ordering=ordering, **kwargs)

    class InjectedCholeskySolver(thp.HipCholeskySolver):
        def __init__(self, objective, linearization_cls=None, linearization_kwargs=None, **kwargs):
            if linearization_cls is None or linearization_cls is ref_dense:
                linearization_cls = InjectedLinearization
            super().__init__(objective, linearization_cls=linearization_cls, linearization_kwargs=linearization_kwargs, **kwargs)

    # th.LUDenseSolver keeps the reference's own DenseLinearization (it is not the class under test); test_theseus_layer.py:183
    # checks ``isinstance(solver.linearization, th.DenseLinearization)`` against the substituted NAME for both dense solvers
    InjectedLinearization.register(ref_dense)
    InjectedLinearization.__name__ = "DenseLinearization"
    InjectedCholeskySolver.__name__ = "CholeskyDenseSolver"
    which = os.environ.get("THX_INJECT", "solver,linearization").split(",")
    if "solver" in which:
        th.CholeskyDenseSolver = tho.CholeskyDenseSolver = thl.CholeskyDenseSolver = InjectedCholeskySolver
        nls.CholeskyDenseSolver = InjectedCholeskySolver
    if "linearization" in which:
        # (dense_solver.DenseLinearization stays the reference's: LUDenseSolver's ``is DenseLinearization`` check, dense_solver.py:28-32)
        th.DenseLinearization = tho.DenseLinearization = InjectedLinearization
    config._thx_injected = which
